// grhip_fir_kernels.h -- the reference's KERNEL-level plug-in seams filled in for gfx950 (SURVEY 8b seams 2 and 3,
// 8a row a14), plus the N-port adapter blocks:
//
//   gr_fir_{ccf,fff,ccc}_hip : gr_fir_{ccf,fff,ccc}     filter / filterN / filterNdec through the C ABI
//                                                       (filter/gr_fir_XXX.h.t:48-122)
//   grhip_fir_sysconfig::get_gr_fir_XXX_info / create   one more `info{name, create}` entry, "hip-gfx950", as every
//                                                       platform's gr_fir_sysconfig adds its own
//                                                       (filter/generate_gr_fir_util.py:25-32,91-115;
//                                                        filter/gr_fir_sysconfig_x86.cc:175-407), so that the
//                                                       reference's "for each implementation" QA and benchmarks
//                                                       (filter/qa_gr_fir_ccf.cc:162-177) cover the GPU path
//   (gr_fft_vcc_hip : gr_fft_vcc + gr_make_fft_vcc_hip, the FFT factory seam, lives in grhip_blocks.h)
//   grhip_stream_to_streams / streams_to_stream / stream_to_vector / vector_to_streams / head blocks
//                                                       (general/gr_stream_to_streams.cc:32-66, gr_streams_to_stream.cc,
//                                                        gr_stream_to_vector.cc, gr_vector_to_streams.cc, gr_head.cc)
// A single filter() call costs a kernel launch: the per-sample loops of the reference (xlating, PFB) must not be
// pointed at this class -- the block-level wrappers (grhip_blocks.h) are the fast path; this class is what makes
// gr_fir_filter_XXX (which does call filterN / filterNdec, filter/gr_fir_filter_XXX.cc.t:81-85) and the QA work.
#pragma once
#include "grhip_blocks.h"

#define GRHIP_FIR_IMPL(NAME, BASE, KIND, I, O, TAP)                                                              \
    class NAME : public BASE {                                                                                   \
        grhip_fir_filter *d_h = nullptr;                                                                         \
        int d_device;                                                                                            \
        void rebuild()                                                                                           \
        {                                                                                                        \
            if (d_h) grhip_fir_filter_destroy(d_h);                                                              \
            d_h = nullptr;                                                                                       \
            const std::vector<TAP> fwd = get_taps();                                                             \
            grhip_detail::check(grhip_fir_filter_create(&d_h, KIND, 1, (const float *)fwd.data(), fwd.size(),    \
                                                        d_device));                                              \
        }                                                                                                        \
    public:                                                                                                      \
        NAME(const std::vector<TAP> &taps, int device = 0) : BASE(taps), d_device(device) { rebuild(); }         \
        ~NAME() { if (d_h) grhip_fir_filter_destroy(d_h); }                                                      \
        O filter(const I input[]) override { O y = O(); filterNdec(&y, input, 1, 1); return y; }                 \
        void filterN(O output[], const I input[], unsigned long n) override { filterNdec(output, input, n, 1); } \
        void filterNdec(O output[], const I input[], unsigned long n, unsigned decimate) override                \
        {                                                                                                        \
            if (n == 0) return;                                                                                  \
            if (ntaps() == 0) { for (unsigned long i = 0; i < n; ++i) output[i] = O(); return; }                 \
            grhip_detail::check(grhip_fir_filterNdec(d_h, output, input, n, decimate));                          \
        }                                                                                                        \
        void set_taps(const std::vector<TAP> &taps) override { BASE::set_taps(taps); rebuild(); }                \
        /* numeric mode of the underlying handle (GRHIP_MODE_*) */                                               \
        void set_mode(int mode) { grhip_detail::check(grhip_fir_filter_set_mode(d_h, mode)); }                   \
    };
GRHIP_FIR_IMPL(gr_fir_ccf_hip, gr_fir_ccf, "ccf", gr_complex, gr_complex, float)
GRHIP_FIR_IMPL(gr_fir_fff_hip, gr_fir_fff, "fff", float, float, float)
GRHIP_FIR_IMPL(gr_fir_ccc_hip, gr_fir_ccc, "ccc", gr_complex, gr_complex, gr_complex)

// what a platform's gr_fir_sysconfig does for its implementations (filter/gr_fir_sysconfig_x86.cc:175-201, 260-300):
// create_gr_fir_XXX picks the best one, get_gr_fir_XXX_info APPENDS to the table the base class started
struct grhip_fir_sysconfig {
    static gr_fir_ccf *create_gr_fir_ccf(const std::vector<float> &taps) { return new gr_fir_ccf_hip(taps); }
    static gr_fir_fff *create_gr_fir_fff(const std::vector<float> &taps) { return new gr_fir_fff_hip(taps); }
    static gr_fir_ccc *create_gr_fir_ccc(const std::vector<gr_complex> &taps) { return new gr_fir_ccc_hip(taps); }
    static void get_gr_fir_ccf_info(std::vector<gr_fir_ccf_info> *info)
    {
        gr_fir_ccf_info t; t.name = "hip-gfx950"; t.create = create_gr_fir_ccf; info->push_back(t);
    }
    static void get_gr_fir_fff_info(std::vector<gr_fir_fff_info> *info)
    {
        gr_fir_fff_info t; t.name = "hip-gfx950"; t.create = create_gr_fir_fff; info->push_back(t);
    }
    static void get_gr_fir_ccc_info(std::vector<gr_fir_ccc_info> *info)
    {
        gr_fir_ccc_info t; t.name = "hip-gfx950"; t.create = create_gr_fir_ccc; info->push_back(t);
    }
};

// ---------------------------------------------------------------------------
// N-port adapters
// ---------------------------------------------------------------------------
class grhip_stream_to_streams_blk : public gr_sync_decimator {       // general/gr_stream_to_streams.cc:32-66
    grhip_stream_adapter *d_h = nullptr;
public:
    grhip_stream_to_streams_blk(size_t item_size, size_t nstreams, int device = 0)
        : gr_sync_decimator("stream_to_streams", gr_make_io_signature(1, 1, item_size),
                            gr_make_io_signature(nstreams, nstreams, item_size), nstreams)
    {
        grhip_detail::check(grhip_stream_adapter_create(&d_h, 1, item_size, nstreams, device));
    }
    ~grhip_stream_to_streams_blk() { grhip_stream_adapter_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        grhip_detail::check(grhip_stream_adapter_work(d_h, noutput_items, const_cast<void *>(in[0]), out.data()));
        return noutput_items;
    }
};
class grhip_streams_to_stream_blk : public gr_sync_interpolator {    // general/gr_streams_to_stream.cc:32-69
    grhip_stream_adapter *d_h = nullptr;
    size_t d_n;
public:
    grhip_streams_to_stream_blk(size_t item_size, size_t nstreams, int device = 0)
        : gr_sync_interpolator("streams_to_stream", gr_make_io_signature(nstreams, nstreams, item_size),
                               gr_make_io_signature(1, 1, item_size), nstreams), d_n(nstreams)
    {
        grhip_detail::check(grhip_stream_adapter_create(&d_h, 0, item_size, nstreams, device));
    }
    ~grhip_streams_to_stream_blk() { grhip_stream_adapter_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        std::vector<void *> ins(in.size());
        for (size_t j = 0; j < in.size(); ++j) ins[j] = const_cast<void *>(in[j]);
        grhip_detail::check(grhip_stream_adapter_work(d_h, noutput_items / (int)d_n, out[0], ins.data()));   // .cc:56-57
        return noutput_items;
    }
};
class grhip_vector_to_streams_blk : public gr_sync_block {           // general/gr_vector_to_streams.cc:31-70
    grhip_stream_adapter *d_h = nullptr;
public:
    grhip_vector_to_streams_blk(size_t item_size, size_t nstreams, int device = 0)
        : gr_sync_block("vector_to_streams", gr_make_io_signature(1, 1, nstreams * item_size),
                        gr_make_io_signature(nstreams, nstreams, item_size))
    {
        grhip_detail::check(grhip_stream_adapter_create(&d_h, 1, item_size, nstreams, device));
    }
    ~grhip_vector_to_streams_blk() { grhip_stream_adapter_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        grhip_detail::check(grhip_stream_adapter_work(d_h, noutput_items, const_cast<void *>(in[0]), out.data()));
        return noutput_items;
    }
};
class grhip_stream_to_vector_blk : public gr_sync_decimator {        // general/gr_stream_to_vector.cc:31-60
    grhip_copy_adapter *d_h = nullptr;
public:
    grhip_stream_to_vector_blk(size_t item_size, size_t nitems_per_block, int device = 0)
        : gr_sync_decimator("stream_to_vector", gr_make_io_signature(1, 1, item_size),
                            gr_make_io_signature(1, 1, item_size * nitems_per_block), nitems_per_block)
    {
        grhip_detail::check(grhip_stream_to_vector_create(&d_h, item_size, nitems_per_block, device));
    }
    ~grhip_stream_to_vector_blk() { grhip_copy_adapter_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_copy_adapter_work(d_h, noutput_items, in[0], out[0]);
        grhip_detail::check(r);
        return r;
    }
};
class grhip_head_blk : public gr_sync_block {                        // general/gr_head.cc:31-62
    grhip_copy_adapter *d_h = nullptr;
public:
    grhip_head_blk(size_t sizeof_stream_item, unsigned long long nitems, int device = 0)
        : gr_sync_block("head", gr_make_io_signature(1, 1, sizeof_stream_item), gr_make_io_signature(1, 1, sizeof_stream_item))
    {
        grhip_detail::check(grhip_head_create(&d_h, sizeof_stream_item, nitems, device));
    }
    ~grhip_head_blk() { grhip_copy_adapter_destroy(d_h); }
    void reset() { grhip_detail::check(grhip_head_reset(d_h)); }       // gr_head.h:51
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        int r = grhip_copy_adapter_work(d_h, noutput_items, in[0], out[0]);
        grhip_detail::check(r);
        return r == GRHIP_WORK_DONE ? -1 : r;                           // WORK_DONE is -1 in gr_block.h:63-66
    }
};
template <class B, class... A> inline boost::shared_ptr<B> grhip_make_adapter(A... a) { return gnuradio::get_initial_sptr(new B(a...)); }
