// gr_shim.h -- the small part of the GNU Radio 3.5 block interface that a signal
// processing block touches, so that the grhip block wrappers (grhip_blocks.h)
// compile and run without Boost / the GNU Radio runtime.  Same names, argument
// meaning and return conventions as the reference:
//   gr_block            gnuradio-core/src/lib/runtime/gr_block.h:63-66,76-84,107-127,153-182
//   gr_sync_block       gnuradio-core/src/lib/runtime/gr_sync_block.cc:38-68
//   gr_sync_decimator   gnuradio-core/src/lib/runtime/gr_sync_decimator.cc:38-68
//   gr_sync_interpolator gnuradio-core/src/lib/runtime/gr_sync_interpolator.cc:30-75
//   gr_io_signature     gnuradio-core/src/lib/runtime/gr_io_signature.h
// When building against a real GNU Radio 3.5 tree define GRHIP_USE_GNURADIO and
// the real headers are used instead (the wrappers only rely on what is here).
#pragma once

#ifdef GRHIP_USE_GNURADIO
#include <gr_block.h>
#include <gr_io_signature.h>
#include <gr_sync_block.h>
#include <gr_sync_decimator.h>
#include <gr_sync_interpolator.h>
#else

#include <cmath>
#include <complex>
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

#endif  // GRHIP_USE_GNURADIO
