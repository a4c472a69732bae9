// grhip_blocks.h -- drop-in gr_sync_block / gr_sync_decimator / gr_block subclasses
// over the C ABI of libgrhip.so.  One class per reference block, with the
// reference's factory signature, io_signature, history, relative rate, setter
// names and exception types:
//
//   grhip_fir_filter_ccf / _fff / _ccc      <- gr_fir_filter_XXX   (filter/gr_fir_filter_XXX.h.t:36-66)
//   grhip_freq_xlating_fir_filter_ccc        <- gr_freq_xlating_fir_filter_ccc (.h.t:64-99)
//   grhip_quadrature_demod_cf                <- gr_quadrature_demod_cf (general/gr_quadrature_demod_cf.h)
//   grhip_clock_recovery_mm_ff               <- digital_clock_recovery_mm_ff (gr-digital/include/...h:44-92)
//   grhip_binary_slicer_fb                   <- digital_binary_slicer_fb
//   grhip_correlate_access_code_bb           <- digital_correlate_access_code_bb
//   gr_fft_vcc_hip (grhip_make_fft_vcc)       <- gr_fft_vcc_fftw, on the abstract gr_fft_vcc base (general/gr_fft_vcc.h:41-59)
//   grhip_pfb_channelizer_ccf                <- gr_pfb_channelizer_ccf (filter/gr_pfb_channelizer_ccf.h:115-178)
//
// output_multiple is the REFERENCE's for every block (1; nsamples for fft_filter_ccc; the
// channeliser's own), so a finite flowgraph produces exactly the items the reference block
// produces, tail included.  The scheduler then hands a block at most half a 64 KiB buffer per
// call (runtime/gr_block_executor.cc:76-78, runtime/gr_flat_flowgraph.cc:37,100): correct, but
// launch-bound on a GPU.  An application that streams can opt in to larger calls
// with grhip_set_batch_items(block, n) below: it multiplies output_multiple, which GNU Radio
// 3.5 honours when it sizes buffers (runtime/gr_flat_flowgraph.cc:102-104,118) -- at the
// documented price of every raised output_multiple in GNU Radio: when the upstream finishes,
// fewer than one multiple of outputs can no longer be requested and that tail is dropped
// (runtime/gr_block_executor.cc:335-348).
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/grhip.h"
#include "gr_shim.h"

namespace grhip_detail {
// status -> the exception type the reference throws for the same precondition
inline void check(int rc)
{
    if (rc >= 0) return;
    std::string msg = std::string(grhip_strerror(rc)) + ": " + grhip_last_error();
    switch (rc) {
    case GRHIP_EINVAL: throw std::invalid_argument(msg);
    case GRHIP_ERANGE: throw std::out_of_range(msg);
    case GRHIP_ENOMEM: throw std::bad_alloc();
    default: throw std::runtime_error(msg);
    }
}
}  // namespace grhip_detail

// opt-in batching (see the header comment): work() calls of n times the block's own output multiple
template <class BLOCK_SPTR> inline void grhip_set_batch_items(const BLOCK_SPTR &b, int n)
{
    if (n < 1) throw std::invalid_argument("grhip_set_batch_items: n must be >= 1");
    b->set_output_multiple(b->output_multiple() * n);
}

// ---------------------------------------------------------------------------
// gr_fir_filter_XXX
// ---------------------------------------------------------------------------
template <class IN, class OUT, class TAP> class grhip_fir_filter_base : public gr_sync_decimator {
protected:
    grhip_fir_filter *d_h = nullptr;
    grhip_fir_filter_base(const char *name, const char *kind, int decimation, const std::vector<TAP> &taps,
                          int device)
        : gr_sync_decimator(name, gr_make_io_signature(1, 1, sizeof(IN)), gr_make_io_signature(1, 1, sizeof(OUT)),
                            decimation)
    {
        grhip_detail::check(grhip_fir_filter_create(&d_h, kind, decimation, (const float *)taps.data(), taps.size(),
                                                    device));
        set_history(grhip_fir_filter_history(d_h));          // set_history(d_fir->ntaps()), .cc.t:51
    }
public:
    ~grhip_fir_filter_base() { grhip_fir_filter_destroy(d_h); }
    void set_taps(const std::vector<TAP> &taps)                // .cc.t:59-64
    {
        grhip_detail::check(grhip_fir_filter_set_taps(d_h, (const float *)taps.data(), taps.size()));
    }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_fir_filter_work(d_h, noutput_items, in[0], out[0]);
        grhip_detail::check(r);
        if (r == 0) set_history(grhip_fir_filter_history(d_h));   // taps changed: history may have too
        return r;
    }
};

#define GRHIP_FIR_CLASS(NAME, KIND, IN, OUT, TAP)                                                          \
    class NAME;                                                                                            \
    typedef boost::shared_ptr<NAME> NAME##_sptr;                                                           \
    NAME##_sptr grhip_make_##KIND(int decimation, const std::vector<TAP> &taps, int device);               \
    class NAME : public grhip_fir_filter_base<IN, OUT, TAP> {                                              \
        friend NAME##_sptr grhip_make_##KIND(int, const std::vector<TAP> &, int);                          \
        NAME(int decimation, const std::vector<TAP> &taps, int device)                                     \
            : grhip_fir_filter_base<IN, OUT, TAP>(#KIND, &#KIND[11], decimation, taps, device) {}          \
    };                                                                                                     \
    inline NAME##_sptr grhip_make_##KIND(int decimation, const std::vector<TAP> &taps, int device = 0)     \
    {                                                                                                      \
        return gnuradio::get_initial_sptr(new NAME(decimation, taps, device));                             \
    }
// &"fir_filter_ccf"[11] == "ccf"
GRHIP_FIR_CLASS(grhip_fir_filter_ccf, fir_filter_ccf, gr_complex, gr_complex, float)
GRHIP_FIR_CLASS(grhip_fir_filter_fff, fir_filter_fff, float, float, float)
GRHIP_FIR_CLASS(grhip_fir_filter_ccc, fir_filter_ccc, gr_complex, gr_complex, gr_complex)

// ---------------------------------------------------------------------------
// gr_freq_xlating_fir_filter_ccc
// ---------------------------------------------------------------------------
class grhip_freq_xlating_fir_filter_ccc_blk;
typedef boost::shared_ptr<grhip_freq_xlating_fir_filter_ccc_blk> grhip_freq_xlating_fir_filter_ccc_sptr;
class grhip_freq_xlating_fir_filter_ccc_blk : public gr_sync_decimator {
    grhip_freq_xlating_fir_filter_ccc *d_h = nullptr;
    grhip_freq_xlating_fir_filter_ccc_blk(int decimation, const std::vector<gr_complex> &taps, double center_freq,
                                          double sampling_freq, int device)
        : gr_sync_decimator("freq_xlating_fir_filter_ccc", gr_make_io_signature(1, 1, sizeof(gr_complex)),
                            gr_make_io_signature(1, 1, sizeof(gr_complex)), decimation)
    {
        grhip_detail::check(grhip_freq_xlating_fir_filter_ccc_create(&d_h, decimation, (const float *)taps.data(),
                                                                     taps.size(), center_freq, sampling_freq, device));
        set_history(grhip_freq_xlating_fir_filter_ccc_history(d_h));
    }
    friend grhip_freq_xlating_fir_filter_ccc_sptr grhip_make_freq_xlating_fir_filter_ccc(
        int, const std::vector<gr_complex> &, double, double, int);
public:
    ~grhip_freq_xlating_fir_filter_ccc_blk() { grhip_freq_xlating_fir_filter_ccc_destroy(d_h); }
    void set_center_freq(double f) { grhip_detail::check(grhip_freq_xlating_fir_filter_ccc_set_center_freq(d_h, f)); }
    void set_taps(const std::vector<gr_complex> &taps)
    {
        grhip_detail::check(grhip_freq_xlating_fir_filter_ccc_set_taps(d_h, (const float *)taps.data(), taps.size()));
    }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_freq_xlating_fir_filter_ccc_work(d_h, noutput_items, in[0], out[0]);
        grhip_detail::check(r);
        if (r == 0) set_history(grhip_freq_xlating_fir_filter_ccc_history(d_h));
        return r;
    }
};
inline grhip_freq_xlating_fir_filter_ccc_sptr grhip_make_freq_xlating_fir_filter_ccc(
    int decimation, const std::vector<gr_complex> &taps, double center_freq, double sampling_freq, int device = 0)
{
    return gnuradio::get_initial_sptr(
        new grhip_freq_xlating_fir_filter_ccc_blk(decimation, taps, center_freq, sampling_freq, device));
}

// ---------------------------------------------------------------------------
// gr_quadrature_demod_cf
// ---------------------------------------------------------------------------
class grhip_quadrature_demod_cf_blk;
typedef boost::shared_ptr<grhip_quadrature_demod_cf_blk> grhip_quadrature_demod_cf_sptr;
class grhip_quadrature_demod_cf_blk : public gr_sync_block {
    grhip_quadrature_demod_cf *d_h = nullptr;
    grhip_quadrature_demod_cf_blk(float gain, int device)
        : gr_sync_block("quadrature_demod_cf", gr_make_io_signature(1, 1, sizeof(gr_complex)),
                        gr_make_io_signature(1, 1, sizeof(float)))
    {
        grhip_detail::check(grhip_quadrature_demod_cf_create(&d_h, gain, device));
        set_history(2);                                        // gr_quadrature_demod_cf.cc:37
    }
    friend grhip_quadrature_demod_cf_sptr grhip_make_quadrature_demod_cf(float, int);
public:
    ~grhip_quadrature_demod_cf_blk() { grhip_quadrature_demod_cf_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_quadrature_demod_cf_work(d_h, noutput_items, in[0], out[0]);
        grhip_detail::check(r);
        return r;
    }
};
inline grhip_quadrature_demod_cf_sptr grhip_make_quadrature_demod_cf(float gain, int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_quadrature_demod_cf_blk(gain, device));
}

// ---------------------------------------------------------------------------
// digital_clock_recovery_mm_ff  (a gr_block: general_work + forecast + consume_each)
// ---------------------------------------------------------------------------
class grhip_clock_recovery_mm_ff_blk;
typedef boost::shared_ptr<grhip_clock_recovery_mm_ff_blk> grhip_clock_recovery_mm_ff_sptr;
class grhip_clock_recovery_mm_ff_blk : public gr_block {
    grhip_clock_recovery_mm_ff *d_h = nullptr;
    grhip_clock_recovery_mm_ff_blk(float omega, float gain_omega, float mu, float gain_mu,
                                   float omega_relative_limit, int device)
        : gr_block("clock_recovery_mm_ff", gr_make_io_signature(1, 1, sizeof(float)),
                   gr_make_io_signature(1, 1, sizeof(float)))
    {
        grhip_detail::check(grhip_clock_recovery_mm_ff_create(&d_h, omega, gain_omega, mu, gain_mu,
                                                              omega_relative_limit, device));
        set_relative_rate(1.0 / omega);                        // .cc:64
    }
    friend grhip_clock_recovery_mm_ff_sptr grhip_make_clock_recovery_mm_ff(float, float, float, float, float, int);
public:
    ~grhip_clock_recovery_mm_ff_blk() { grhip_clock_recovery_mm_ff_destroy(d_h); }
    void forecast(int noutput_items, gr_vector_int &req) override
    {
        int n = grhip_clock_recovery_mm_ff_forecast(d_h, noutput_items);
        grhip_detail::check(n);
        for (size_t i = 0; i < req.size(); i++) req[i] = n;
    }
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &in,
                     gr_vector_void_star &out) override
    {
        int consumed = 0;
        int r = grhip_clock_recovery_mm_ff_general_work(d_h, noutput_items, ninput_items[0], (const float *)in[0],
                                                        (float *)out[0], &consumed);
        grhip_detail::check(r);
        consume_each(consumed);
        return r;
    }
    float mu() const { return grhip_clock_recovery_mm_ff_mu(d_h); }
    float omega() const { return grhip_clock_recovery_mm_ff_omega(d_h); }
    float gain_mu() const { return grhip_clock_recovery_mm_ff_gain_mu(d_h); }
    float gain_omega() const { return grhip_clock_recovery_mm_ff_gain_omega(d_h); }
    void set_gain_mu(float v) { grhip_detail::check(grhip_clock_recovery_mm_ff_set_gain_mu(d_h, v)); }
    void set_gain_omega(float v) { grhip_detail::check(grhip_clock_recovery_mm_ff_set_gain_omega(d_h, v)); }
    void set_mu(float v) { grhip_detail::check(grhip_clock_recovery_mm_ff_set_mu(d_h, v)); }
    void set_omega(float v) { grhip_detail::check(grhip_clock_recovery_mm_ff_set_omega(d_h, v)); }
};
inline grhip_clock_recovery_mm_ff_sptr grhip_make_clock_recovery_mm_ff(float omega, float gain_omega, float mu,
                                                                       float gain_mu, float omega_relative_limit,
                                                                       int device = 0)
{
    return gnuradio::get_initial_sptr(
        new grhip_clock_recovery_mm_ff_blk(omega, gain_omega, mu, gain_mu, omega_relative_limit, device));
}

// ---------------------------------------------------------------------------
// digital_binary_slicer_fb, digital_correlate_access_code_bb
// ---------------------------------------------------------------------------
class grhip_binary_slicer_fb_blk;
typedef boost::shared_ptr<grhip_binary_slicer_fb_blk> grhip_binary_slicer_fb_sptr;
class grhip_binary_slicer_fb_blk : public gr_sync_block {
    grhip_binary_slicer_fb *d_h = nullptr;
    explicit grhip_binary_slicer_fb_blk(int device)
        : gr_sync_block("binary_slicer_fb", gr_make_io_signature(1, 1, sizeof(float)),
                        gr_make_io_signature(1, 1, sizeof(unsigned char)))
    {
        grhip_detail::check(grhip_binary_slicer_fb_create(&d_h, device));
    }
    friend grhip_binary_slicer_fb_sptr grhip_make_binary_slicer_fb(int);
public:
    ~grhip_binary_slicer_fb_blk() { grhip_binary_slicer_fb_destroy(d_h); }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_binary_slicer_fb_work(d_h, n, (const float *)in[0], (unsigned char *)out[0]);
        grhip_detail::check(r);
        return r;
    }
};
inline grhip_binary_slicer_fb_sptr grhip_make_binary_slicer_fb(int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_binary_slicer_fb_blk(device));
}

// digital_clock_recovery_mm_cc (gr-digital/include/digital_clock_recovery_mm_cc.h:44-110): one complex input,
// one complex output and the optional float error output (the shim's io signature carries one item size, so the
// second port is described by the comment only; general_work looks at out.size() like the reference)
class grhip_clock_recovery_mm_cc_blk;
typedef boost::shared_ptr<grhip_clock_recovery_mm_cc_blk> grhip_clock_recovery_mm_cc_sptr;
class grhip_clock_recovery_mm_cc_blk : public gr_block {
    grhip_clock_recovery_mm_cc *d_h = nullptr;
    grhip_clock_recovery_mm_cc_blk(float omega, float gain_omega, float mu, float gain_mu, float omega_relative_limit,
                                   int device)
        : gr_block("clock_recovery_mm_cc", gr_make_io_signature(1, 1, sizeof(gr_complex)),
                   gr_make_io_signature(1, 2, sizeof(gr_complex)))
    {
        grhip_detail::check(grhip_clock_recovery_mm_cc_create(&d_h, omega, gain_omega, mu, gain_mu,
                                                              omega_relative_limit, device));
        set_relative_rate(1.0 / omega);                        // .cc:68
        set_history(3);                                        // .cc:69
    }
    friend grhip_clock_recovery_mm_cc_sptr grhip_make_clock_recovery_mm_cc(float, float, float, float, float, int);
    float get(int (*f)(grhip_clock_recovery_mm_cc *, float *)) const
    {
        float v = 0;
        grhip_detail::check(f(d_h, &v));
        return v;
    }
public:
    ~grhip_clock_recovery_mm_cc_blk() { grhip_clock_recovery_mm_cc_destroy(d_h); }
    void forecast(int noutput_items, gr_vector_int &req) override
    {
        int n = grhip_clock_recovery_mm_cc_forecast(d_h, noutput_items);
        grhip_detail::check(n);
        for (size_t i = 0; i < req.size(); i++) req[i] = n;
    }
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &in,
                     gr_vector_void_star &out) override
    {
        int consumed = 0;
        float *err = out.size() >= 2 ? (float *)out[1] : nullptr;          // .cc:124-126
        int r = grhip_clock_recovery_mm_cc_general_work(d_h, noutput_items, ninput_items[0], in[0], out[0], err, &consumed);
        grhip_detail::check(r);
        consume_each(consumed);
        return r;
    }
    float mu() const { return get(grhip_clock_recovery_mm_cc_mu); }
    float omega() const { return get(grhip_clock_recovery_mm_cc_omega); }
    float gain_mu() const { return get(grhip_clock_recovery_mm_cc_gain_mu); }
    float gain_omega() const { return get(grhip_clock_recovery_mm_cc_gain_omega); }
    void set_gain_mu(float v) { grhip_detail::check(grhip_clock_recovery_mm_cc_set_gain_mu(d_h, v)); }
    void set_gain_omega(float v) { grhip_detail::check(grhip_clock_recovery_mm_cc_set_gain_omega(d_h, v)); }
    void set_mu(float v) { grhip_detail::check(grhip_clock_recovery_mm_cc_set_mu(d_h, v)); }
    void set_omega(float v) { grhip_detail::check(grhip_clock_recovery_mm_cc_set_omega(d_h, v)); }
};
inline grhip_clock_recovery_mm_cc_sptr grhip_make_clock_recovery_mm_cc(float omega, float gain_omega, float mu,
                                                                       float gain_mu, float omega_relative_limit,
                                                                       int device = 0)
{
    return gnuradio::get_initial_sptr(
        new grhip_clock_recovery_mm_cc_blk(omega, gain_omega, mu, gain_mu, omega_relative_limit, device));
}

// pager_slicer_fb (gr-pager/lib/pager_slicer_fb.h:30-58), gr_unpack_k_bits_bb (general/gr_unpack_k_bits_bb.h)
class grhip_pager_slicer_fb_blk;
typedef boost::shared_ptr<grhip_pager_slicer_fb_blk> grhip_pager_slicer_fb_sptr;
class grhip_pager_slicer_fb_blk : public gr_sync_block {
    grhip_pager_slicer_fb *d_h = nullptr;
    grhip_pager_slicer_fb_blk(float alpha, int device)
        : gr_sync_block("slicer_fb", gr_make_io_signature(1, 1, sizeof(float)),
                        gr_make_io_signature(1, 1, sizeof(unsigned char)))
    {
        grhip_detail::check(grhip_pager_slicer_fb_create(&d_h, alpha, device));
    }
    friend grhip_pager_slicer_fb_sptr grhip_make_pager_slicer_fb(float, int);
public:
    ~grhip_pager_slicer_fb_blk() { grhip_pager_slicer_fb_destroy(d_h); }
    float dc_offset() const
    {
        float v = 0;
        grhip_detail::check(grhip_pager_slicer_fb_dc_offset(d_h, &v));
        return v;
    }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_pager_slicer_fb_work(d_h, n, (const float *)in[0], (unsigned char *)out[0]);
        grhip_detail::check(r);
        return r;
    }
};
inline grhip_pager_slicer_fb_sptr grhip_make_pager_slicer_fb(float alpha, int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_pager_slicer_fb_blk(alpha, device));
}

class grhip_unpack_k_bits_bb_blk;
typedef boost::shared_ptr<grhip_unpack_k_bits_bb_blk> grhip_unpack_k_bits_bb_sptr;
class grhip_unpack_k_bits_bb_blk : public gr_sync_interpolator {
    grhip_unpack_k_bits_bb *d_h = nullptr;
    grhip_unpack_k_bits_bb_blk(unsigned k, int device)
        : gr_sync_interpolator("unpack_k_bits_bb", gr_make_io_signature(1, 1, sizeof(unsigned char)),
                               gr_make_io_signature(1, 1, sizeof(unsigned char)), k)
    {
        // the reference throws std::out_of_range("interpolation must be > 0") (.cc:45-46)
        grhip_detail::check(grhip_unpack_k_bits_bb_create(&d_h, k, device));
    }
    friend grhip_unpack_k_bits_bb_sptr grhip_make_unpack_k_bits_bb(unsigned, int);
public:
    ~grhip_unpack_k_bits_bb_blk() { grhip_unpack_k_bits_bb_destroy(d_h); }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_unpack_k_bits_bb_work(d_h, n, (const unsigned char *)in[0], (unsigned char *)out[0]);
        grhip_detail::check(r);
        return r;
    }
};
inline grhip_unpack_k_bits_bb_sptr grhip_make_unpack_k_bits_bb(unsigned k, int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_unpack_k_bits_bb_blk(k, device));
}

// gr_framer_sink_1 (general/gr_framer_sink_1.h:34-107): same constructor argument, same messages in the same queue
class grhip_framer_sink_1_blk;
typedef boost::shared_ptr<grhip_framer_sink_1_blk> grhip_framer_sink_1_sptr;
class grhip_framer_sink_1_blk : public gr_sync_block {
    grhip_framer_sink_1 *d_h = nullptr;
    gr_msg_queue_sptr d_target_queue;
    std::vector<unsigned char> d_buf;
    grhip_framer_sink_1_blk(gr_msg_queue_sptr target_queue, int device)
        : gr_sync_block("framer_sink_1", gr_make_io_signature(1, 1, sizeof(unsigned char)), gr_make_io_signature(0, 0, 0)),
          d_target_queue(target_queue), d_buf(4096)
    {
        grhip_detail::check(grhip_framer_sink_1_create(&d_h, device));
    }
    friend grhip_framer_sink_1_sptr grhip_make_framer_sink_1(gr_msg_queue_sptr, int);
public:
    ~grhip_framer_sink_1_blk() { grhip_framer_sink_1_destroy(d_h); }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &) override
    {
        int r = grhip_framer_sink_1_work(d_h, n, (const unsigned char *)in[0]);
        grhip_detail::check(r);
        int m = grhip_framer_sink_1_message_count(d_h, nullptr);
        grhip_detail::check(m);
        for (int i = 0; i < m; i++) {
            int woff = 0;
            int len = grhip_framer_sink_1_pop(d_h, &woff, d_buf.data(), (int)d_buf.size());
            grhip_detail::check(len);
            gr_message_sptr msg = gr_make_message(0, woff, 0, len);      // .cc:140-141, 168-170
            if (len) memcpy(msg->msg(), d_buf.data(), len);
            d_target_queue->insert_tail(msg);
        }
        return r;
    }
};
inline grhip_framer_sink_1_sptr grhip_make_framer_sink_1(gr_msg_queue_sptr target_queue, int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_framer_sink_1_blk(target_queue, device));
}

class grhip_correlate_access_code_bb_blk;
typedef boost::shared_ptr<grhip_correlate_access_code_bb_blk> grhip_correlate_access_code_bb_sptr;
class grhip_correlate_access_code_bb_blk : public gr_sync_block {
    grhip_correlate_access_code_bb *d_h = nullptr;
    grhip_correlate_access_code_bb_blk(const std::string &access_code, int threshold, int device)
        : gr_sync_block("correlate_access_code_bb", gr_make_io_signature(1, 1, sizeof(char)),
                        gr_make_io_signature(1, 1, sizeof(char)))
    {
        // the reference throws std::out_of_range("access_code is > 64 bits") (.cc:54-57)
        grhip_detail::check(grhip_correlate_access_code_bb_create(&d_h, access_code.data(), access_code.size(),
                                                                  threshold, device));
    }
    friend grhip_correlate_access_code_bb_sptr grhip_make_correlate_access_code_bb(const std::string &, int, int);
public:
    ~grhip_correlate_access_code_bb_blk() { grhip_correlate_access_code_bb_destroy(d_h); }
    bool set_access_code(const std::string &code)             // .cc:64-85: false if longer than 64
    {
        return grhip_correlate_access_code_bb_set_access_code(d_h, code.data(), code.size()) == GRHIP_OK;
    }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_correlate_access_code_bb_work(d_h, n, (const unsigned char *)in[0], (unsigned char *)out[0]);
        grhip_detail::check(r);
        return r;
    }
};
inline grhip_correlate_access_code_bb_sptr grhip_make_correlate_access_code_bb(const std::string &access_code,
                                                                               int threshold, int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_correlate_access_code_bb_blk(access_code, threshold, device));
}

// ---------------------------------------------------------------------------
// gr_fft_vcc_hip: sits where gr_fft_vcc_fftw sits (general/gr_fft_vcc_fftw.h:36-58).  The base class owns size, window,
// direction and shift; its set_window() is not virtual and only stores the vector, so work() hands a changed
// window to the device before it transforms (the FFTW subclass reads d_window in work() too, .cc:68-76).
// ---------------------------------------------------------------------------
class gr_fft_vcc_hip;
typedef boost::shared_ptr<gr_fft_vcc_hip> gr_fft_vcc_hip_sptr;
class gr_fft_vcc_hip : public gr_fft_vcc {
    grhip_fft_vcc *d_h = nullptr;
    std::vector<float> d_sent;
    gr_fft_vcc_hip(int fft_size, bool forward, const std::vector<float> &window, bool shift, int device)
        : gr_fft_vcc("fft_vcc_hip", fft_size, forward, window, shift), d_sent(d_window)
    {
        grhip_detail::check(grhip_fft_vcc_create(&d_h, fft_size, forward, d_window.data(), d_window.size(), shift, device));
    }
    friend gr_fft_vcc_hip_sptr gr_make_fft_vcc_hip(int, bool, const std::vector<float> &, bool, int);
public:
    ~gr_fft_vcc_hip() { grhip_fft_vcc_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        if (d_window != d_sent) {
            grhip_detail::check(grhip_fft_vcc_set_window(d_h, d_window.data(), d_window.size()));
            d_sent = d_window;
        }
        int r = grhip_fft_vcc_work(d_h, noutput_items, in[0], out[0]);
        grhip_detail::check(r);
        return r;
    }
};
inline gr_fft_vcc_hip_sptr gr_make_fft_vcc_hip(int fft_size, bool forward, const std::vector<float> &window,
                                               bool shift = false, int device = 0)
{
    if (fft_size <= 0) throw std::out_of_range("gr_fft_vcc_hip: invalid fft_size");      // gri_fft.cc:104-105
    return gnuradio::get_initial_sptr(new gr_fft_vcc_hip(fft_size, forward, window, shift, device));
}

// the block-level factory of round 1 keeps its name
typedef gr_fft_vcc_hip grhip_fft_vcc_blk;
typedef gr_fft_vcc_hip_sptr grhip_fft_vcc_sptr;
inline grhip_fft_vcc_sptr grhip_make_fft_vcc(int fft_size, bool forward, const std::vector<float> &window,
                                             bool shift = false, int device = 0)
{
    return gr_make_fft_vcc_hip(fft_size, forward, window, shift, device);
}

// ---------------------------------------------------------------------------
// gr_pfb_channelizer_ccf  (numchans inputs, one output of numchans-complex vectors)
// ---------------------------------------------------------------------------
// gr_fft_filter_ccc (filter/gr_fft_filter_ccc.h): gr_sync_decimator, history 1, output multiple nsamples
class grhip_fft_filter_ccc_blk;
typedef boost::shared_ptr<grhip_fft_filter_ccc_blk> grhip_fft_filter_ccc_sptr;
class grhip_fft_filter_ccc_blk : public gr_sync_decimator {
    grhip_fft_filter_ccc *d_h = nullptr;
    grhip_fft_filter_ccc_blk(int decimation, const std::vector<gr_complex> &taps, int device)
        : gr_sync_decimator("fft_filter_ccc", gr_make_io_signature(1, 1, sizeof(gr_complex)),
                            gr_make_io_signature(1, 1, sizeof(gr_complex)), decimation)
    {
        grhip_detail::check(grhip_fft_filter_ccc_create(&d_h, decimation, (const float *)taps.data(), taps.size(), device));
        set_history(1);
        set_output_multiple(grhip_fft_filter_ccc_nsamples(d_h));          // gr_fft_filter_ccc.cc:69
    }
    friend grhip_fft_filter_ccc_sptr grhip_make_fft_filter_ccc(int, const std::vector<gr_complex> &, int);
public:
    ~grhip_fft_filter_ccc_blk() { grhip_fft_filter_ccc_destroy(d_h); }
    void set_taps(const std::vector<gr_complex> &taps)
    {
        grhip_detail::check(grhip_fft_filter_ccc_set_taps(d_h, (const float *)taps.data(), taps.size()));
    }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_fft_filter_ccc_work(d_h, n, in[0], out[0]);
        grhip_detail::check(r);
        if (r == 0) set_output_multiple(grhip_fft_filter_ccc_nsamples(d_h));   // .cc:113-118
        return r;
    }
};
inline grhip_fft_filter_ccc_sptr grhip_make_fft_filter_ccc(int decimation, const std::vector<gr_complex> &taps, int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_fft_filter_ccc_blk(decimation, taps, device));
}

// gr_pfb_decimator_ccf (filter/gr_pfb_decimator_ccf.h:100-140)
class grhip_pfb_decimator_ccf_blk;
typedef boost::shared_ptr<grhip_pfb_decimator_ccf_blk> grhip_pfb_decimator_ccf_sptr;
class grhip_pfb_decimator_ccf_blk : public gr_sync_block {
    grhip_pfb_decimator_ccf *d_h = nullptr;
    grhip_pfb_decimator_ccf_blk(unsigned decim, const std::vector<float> &taps, unsigned channel, int device)
        : gr_sync_block("pfb_decimator_ccf", gr_make_io_signature(decim, decim, sizeof(gr_complex)),
                        gr_make_io_signature(1, 1, sizeof(gr_complex)))
    {
        grhip_detail::check(grhip_pfb_decimator_ccf_create(&d_h, decim, taps.data(), taps.size(), channel, device));
        set_history(grhip_pfb_decimator_ccf_history(d_h));           // .cc:108
    }
    friend grhip_pfb_decimator_ccf_sptr grhip_make_pfb_decimator_ccf(unsigned, const std::vector<float> &, unsigned, int);
public:
    ~grhip_pfb_decimator_ccf_blk() { grhip_pfb_decimator_ccf_destroy(d_h); }
    void set_taps(const std::vector<float> &taps)
    {
        grhip_detail::check(grhip_pfb_decimator_ccf_set_taps(d_h, taps.data(), taps.size()));
        set_history(grhip_pfb_decimator_ccf_history(d_h));
    }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_pfb_decimator_ccf_work(d_h, n, in.data(), out[0]);
        grhip_detail::check(r);
        return r;
    }
};
inline grhip_pfb_decimator_ccf_sptr grhip_make_pfb_decimator_ccf(unsigned decim, const std::vector<float> &taps,
                                                                 unsigned channel = 0, int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_pfb_decimator_ccf_blk(decim, taps, channel, device));
}

class grhip_pfb_channelizer_ccf_blk;
typedef boost::shared_ptr<grhip_pfb_channelizer_ccf_blk> grhip_pfb_channelizer_ccf_sptr;
class grhip_pfb_channelizer_ccf_blk : public gr_block {
    grhip_pfb_channelizer_ccf *d_h = nullptr;
    unsigned d_numchans;
    grhip_pfb_channelizer_ccf_blk(unsigned numchans, const std::vector<float> &taps, float oversample_rate, int device)
        : gr_block("pfb_channelizer_ccf", gr_make_io_signature(numchans, numchans, sizeof(gr_complex)),
                   gr_make_io_signature(1, 1, numchans * sizeof(gr_complex))),
          d_numchans(numchans)
    {
        // std::invalid_argument when numchans/oversample_rate is not an integer (.cc:57-60)
        grhip_detail::check(grhip_pfb_channelizer_ccf_create(&d_h, numchans, taps.data(), taps.size(), oversample_rate,
                                                             device));
        set_history(grhip_pfb_channelizer_ccf_history(d_h));
        set_relative_rate(1.0 / (numchans / oversample_rate));    // set_relative_rate(1.0/intp), .cc:62
        set_output_multiple(grhip_pfb_channelizer_ccf_output_multiple(d_h));     // gr_pfb_channelizer_ccf.cc:92
    }
    friend grhip_pfb_channelizer_ccf_sptr grhip_make_pfb_channelizer_ccf(unsigned, const std::vector<float> &, float, int);
public:
    ~grhip_pfb_channelizer_ccf_blk() { grhip_pfb_channelizer_ccf_destroy(d_h); }
    void set_taps(const std::vector<float> &taps)
    {
        grhip_detail::check(grhip_pfb_channelizer_ccf_set_taps(d_h, taps.data(), taps.size()));
        set_history(grhip_pfb_channelizer_ccf_history(d_h));
    }
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int consumed = 0;
        int r = grhip_pfb_channelizer_ccf_general_work(d_h, noutput_items, in.data(), out[0], &consumed);
        grhip_detail::check(r);
        consume_each(consumed);
        return r;
    }
};
inline grhip_pfb_channelizer_ccf_sptr grhip_make_pfb_channelizer_ccf(unsigned numchans, const std::vector<float> &taps,
                                                                     float oversample_rate = 1, int device = 0)
{
    return gnuradio::get_initial_sptr(new grhip_pfb_channelizer_ccf_blk(numchans, taps, oversample_rate, device));
}
