// gr_shim.h -- the small part of the GNU Radio 3.5 block interface that a signal
// processing block touches, so that the grhip block wrappers (grhip_blocks.h)
// compile and run without Boost / the GNU Radio runtime.  Same names, argument
// meaning and return conventions as the reference:
//   gr_block            gnuradio-core/src/lib/runtime/gr_block.h:63-66,76-84,107-127,153-182
//   gr_sync_block       gnuradio-core/src/lib/runtime/gr_sync_block.cc:38-68
//   gr_sync_decimator   gnuradio-core/src/lib/runtime/gr_sync_decimator.cc:38-68
//   gr_sync_interpolator gnuradio-core/src/lib/runtime/gr_sync_interpolator.cc:30-75
//   gr_io_signature     gnuradio-core/src/lib/runtime/gr_io_signature.h
//   gr_message / gr_msg_queue  gnuradio-core/src/lib/runtime/gr_message.h:36-81, gr_msg_queue.h:36-86
//   gr_fir_{ccf,fff,ccc}, gr_fir_XXX_info   gnuradio-core/src/lib/filter/gr_fir_XXX.h.t:48-122,
//                                           filter/generate_gr_fir_util.py:25-32 (the kernel-level seam)
//   gr_fft_vcc (abstract base)              gnuradio-core/src/lib/general/gr_fft_vcc.h:41-59, gr_fft_vcc.cc:40-64
// When building against a real GNU Radio 3.5 tree define GRHIP_USE_GNURADIO and
// the real headers are used instead (the wrappers only rely on what is here).
#pragma once

#ifdef GRHIP_USE_GNURADIO
#include <gr_fft_vcc.h>
#include <gr_fir_ccc.h>
#include <gr_fir_ccf.h>
#include <gr_fir_fff.h>
#include <gr_fir_util.h>
#include <gr_block.h>
#include <gr_io_signature.h>
#include <gr_message.h>
#include <gr_msg_queue.h>
#include <gr_sync_block.h>
#include <gr_sync_decimator.h>
#include <gr_sync_interpolator.h>
#else

#include <cmath>
#include <complex>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace boost {
// the reference hands blocks around as boost::shared_ptr
template <class T> using shared_ptr = std::shared_ptr<T>;
}
namespace gnuradio {
template <class T> boost::shared_ptr<T> get_initial_sptr(T *p) { return boost::shared_ptr<T>(p); }
}

class gr_io_signature {
    int d_min, d_max;
    std::vector<int> d_sizeof;
public:
    gr_io_signature(int mn, int mx, int size) : d_min(mn), d_max(mx), d_sizeof(1, size) {}
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int) const { return d_sizeof[0]; }
};
typedef boost::shared_ptr<gr_io_signature> gr_io_signature_sptr;
inline gr_io_signature_sptr gr_make_io_signature(int mn, int mx, int size)
{
    return gr_io_signature_sptr(new gr_io_signature(mn, mx, size));
}

class gr_block {
public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };   // gr_block.h:63-66

    virtual ~gr_block() {}
    const std::string &name() const { return d_name; }
    gr_io_signature_sptr input_signature() const { return d_in; }
    gr_io_signature_sptr output_signature() const { return d_out; }

    unsigned history() const { return d_history; }
    void set_history(unsigned h) { d_history = h; }
    int output_multiple() const { return d_output_multiple; }
    void set_output_multiple(int m) { d_output_multiple = m; }
    double relative_rate() const { return d_relative_rate; }
    void set_relative_rate(double r) { d_relative_rate = r; }

    virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required)
    {
        for (size_t i = 0; i < ninput_items_required.size(); i++)
            ninput_items_required[i] = noutput_items + history() - 1;     // gr_block.cc default
    }
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items,
                             gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;

    void consume_each(int n) { d_consumed = n; }
    int consumed() const { return d_consumed; }          // read by the executor after general_work

protected:
    gr_block(const std::string &name, gr_io_signature_sptr in, gr_io_signature_sptr out)
        : d_name(name), d_in(in), d_out(out) {}

private:
    std::string d_name;
    gr_io_signature_sptr d_in, d_out;
    unsigned d_history = 1;
    int d_output_multiple = 1;
    double d_relative_rate = 1.0;
    int d_consumed = 0;
};
typedef boost::shared_ptr<gr_block> gr_block_sptr;

class gr_sync_block : public gr_block {
protected:
    gr_sync_block(const std::string &name, gr_io_signature_sptr in, gr_io_signature_sptr out)
        : gr_block(name, in, out) {}
public:
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) = 0;
    void forecast(int noutput_items, gr_vector_int &req) override
    {
        for (size_t i = 0; i < req.size(); i++) req[i] = noutput_items + history() - 1;   // gr_sync_block.cc:46-50
    }
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &in,
                     gr_vector_void_star &out) override
    {
        int r = work(noutput_items, in, out);
        if (r > 0) consume_each(r); else consume_each(0);              // gr_sync_block.cc:58-66
        return r;
    }
};

class gr_sync_decimator : public gr_sync_block {
    unsigned d_decimation;
protected:
    gr_sync_decimator(const std::string &name, gr_io_signature_sptr in, gr_io_signature_sptr out,
                      unsigned decimation)
        : gr_sync_block(name, in, out), d_decimation(decimation)
    {
        set_relative_rate(1.0 / decimation);
    }
public:
    unsigned decimation() const { return d_decimation; }
    void forecast(int noutput_items, gr_vector_int &req) override
    {
        for (size_t i = 0; i < req.size(); i++)
            req[i] = noutput_items * decimation() + history() - 1;     // gr_sync_decimator.cc:46-50
    }
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &in,
                     gr_vector_void_star &out) override
    {
        int r = work(noutput_items, in, out);
        if (r > 0) consume_each(r * decimation()); else consume_each(0);   // gr_sync_decimator.cc:58-66
        return r;
    }
};

// runtime/gr_sync_interpolator.{h,cc}
class gr_sync_interpolator : public gr_sync_block {
    unsigned d_interpolation;
protected:
    gr_sync_interpolator(const std::string &name, gr_io_signature_sptr in, gr_io_signature_sptr out,
                         unsigned interpolation)
        : gr_sync_block(name, in, out), d_interpolation(interpolation)
    {
        set_relative_rate(1.0 * interpolation);
        set_output_multiple(interpolation);                             // gr_sync_interpolator.cc:34-35
    }
public:
    unsigned interpolation() const { return d_interpolation; }
    void forecast(int noutput_items, gr_vector_int &req) override
    {
        for (size_t i = 0; i < req.size(); i++)
            req[i] = noutput_items / interpolation() + history() - 1;   // gr_sync_interpolator.cc:42-46
    }
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &in,
                     gr_vector_void_star &out) override
    {
        int r = work(noutput_items, in, out);
        if (r > 0) consume_each(r / interpolation()); else consume_each(0);   // gr_sync_interpolator.cc:54-62
        return r;
    }
};

// runtime/gr_message.h: type, two numeric arguments and a byte string
class gr_message {
    long d_type;
    double d_arg1, d_arg2;
    std::vector<unsigned char> d_buf;
public:
    gr_message(long type, double arg1, double arg2, size_t length) : d_type(type), d_arg1(arg1), d_arg2(arg2), d_buf(length) {}
    long type() const { return d_type; }
    double arg1() const { return d_arg1; }
    double arg2() const { return d_arg2; }
    unsigned char *msg() { return d_buf.data(); }
    size_t length() const { return d_buf.size(); }
    std::string to_string() const { return std::string(d_buf.begin(), d_buf.end()); }
};
typedef boost::shared_ptr<gr_message> gr_message_sptr;
inline gr_message_sptr gr_make_message(long type = 0, double arg1 = 0, double arg2 = 0, size_t length = 0)
{
    return gr_message_sptr(new gr_message(type, arg1, arg2, length));
}

// runtime/gr_msg_queue.h: thread-safe FIFO; insert_tail blocks while a bounded queue is full
class gr_msg_queue {
    std::deque<gr_message_sptr> d_q;
    std::mutex d_m;
    std::condition_variable d_not_empty, d_not_full;
    unsigned d_limit;
public:
    explicit gr_msg_queue(unsigned limit = 0) : d_limit(limit) {}
    void insert_tail(gr_message_sptr msg)
    {
        std::unique_lock<std::mutex> l(d_m);
        d_not_full.wait(l, [&] { return d_limit == 0 || d_q.size() < d_limit; });
        d_q.push_back(msg);
        d_not_empty.notify_one();
    }
    gr_message_sptr delete_head()
    {
        std::unique_lock<std::mutex> l(d_m);
        d_not_empty.wait(l, [&] { return !d_q.empty(); });
        gr_message_sptr m = d_q.front();
        d_q.pop_front();
        d_not_full.notify_one();
        return m;
    }
    gr_message_sptr delete_head_nowait()
    {
        std::unique_lock<std::mutex> l(d_m);
        if (d_q.empty()) return gr_message_sptr();
        gr_message_sptr m = d_q.front();
        d_q.pop_front();
        d_not_full.notify_one();
        return m;
    }
    void flush() { while (delete_head_nowait()) {} }
    bool empty_p() { std::unique_lock<std::mutex> l(d_m); return d_q.empty(); }
    bool full_p() { std::unique_lock<std::mutex> l(d_m); return d_limit != 0 && d_q.size() >= d_limit; }
    unsigned count() { std::unique_lock<std::mutex> l(d_m); return (unsigned)d_q.size(); }
    unsigned limit() const { return d_limit; }
};
typedef boost::shared_ptr<gr_msg_queue> gr_msg_queue_sptr;
inline gr_msg_queue_sptr gr_make_msg_queue(unsigned limit = 0) { return gr_msg_queue_sptr(new gr_msg_queue(limit)); }


// ---- kernel-level FIR seam: abstract gr_fir_XXX (filter/gr_fir_XXX.h.t:48-122) -------------------
template <class T> inline std::vector<T> gr_reverse(const std::vector<T> &v) { return std::vector<T>(v.rbegin(), v.rend()); }

#define GRHIP_SHIM_FIR(NAME, I, O, TAP)                                                                     \
    class NAME {                                                                                            \
    protected:                                                                                              \
        std::vector<TAP> d_taps; /* reversed taps */                                                        \
    public:                                                                                                 \
        NAME() {}                                                                                           \
        NAME(const std::vector<TAP> &taps) : d_taps(gr_reverse(taps)) {}                                    \
        virtual ~NAME() {}                                                                                  \
        virtual O filter(const I input[]) = 0;                                                              \
        virtual void filterN(O output[], const I input[], unsigned long n) = 0;                             \
        virtual void filterNdec(O output[], const I input[], unsigned long n, unsigned decimate) = 0;       \
        virtual void set_taps(const std::vector<TAP> &taps) { d_taps = gr_reverse(taps); }                  \
        unsigned ntaps() const { return d_taps.size(); }                                                    \
        virtual const std::vector<TAP> get_taps() const { return gr_reverse(d_taps); }                      \
    };                                                                                                      \
    struct NAME##_info {                                                                                    \
        const char *name; /* implementation name, e.g. "generic", "SSE" */                                  \
        NAME *(*create)(const std::vector<TAP> &taps);                                                      \
    };
GRHIP_SHIM_FIR(gr_fir_ccf, gr_complex, gr_complex, float)
GRHIP_SHIM_FIR(gr_fir_fff, float, float, float)
GRHIP_SHIM_FIR(gr_fir_ccc, gr_complex, gr_complex, gr_complex)

// ---- gr_fft_vcc: the abstract base that holds size / window / direction (general/gr_fft_vcc.h:41-59) ----
class gr_fft_vcc : public gr_sync_block {
protected:
    unsigned int d_fft_size;
    std::vector<float> d_window;
    bool d_forward;
    bool d_shift;
    gr_fft_vcc(const std::string &name, int fft_size, bool forward, const std::vector<float> &window, bool shift)
        : gr_sync_block(name, gr_make_io_signature(1, 1, fft_size * sizeof(gr_complex)),
                        gr_make_io_signature(1, 1, fft_size * sizeof(gr_complex))),
          d_fft_size(fft_size), d_forward(forward), d_shift(shift)
    {
        set_window(window);
    }
public:
    ~gr_fft_vcc() {}
    bool set_window(const std::vector<float> &window)          // gr_fft_vcc.cc:55-64
    {
        if (window.size() == 0 || window.size() == d_fft_size) { d_window = window; return true; }
        return false;
    }
};

#endif  // GRHIP_USE_GNURADIO
