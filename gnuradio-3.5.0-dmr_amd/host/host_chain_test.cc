// host_chain_test.cc -- C++ drop-in test: builds the DMR chain out of the grhip
// block wrappers exactly as a GNU Radio 3.5 C++ application would (factories
// returning shared pointers, connect in order, run), through the stand-in
// executor.  Reads the capture and parameters from files written by the pytest
// that drives it and writes each stage's output for comparison with the oracle.
//
// usage: host_chain_test <dir> <mode: chain|errors>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "grhip_blocks.h"
#include "grhip_executor.h"

template <class T> static std::vector<T> slurp(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    if (!f) { std::cerr << "cannot open " << p << "\n"; exit(2); }
    f.seekg(0, std::ios::end); size_t n = f.tellg(); f.seekg(0);
    std::vector<T> v(n / sizeof(T));
    f.read((char *)v.data(), n);
    return v;
}
static void dump(const std::string &p, const std::vector<unsigned char> &v)
{
    std::ofstream f(p, std::ios::binary);
    f.write((const char *)v.data(), v.size());
}

static int test_errors()
{
    int fails = 0;
    try { grhip_make_clock_recovery_mm_ff(0.5f, 0.01f, 0.5f, 0.01f, 0.001f); fails++; }
    catch (const std::out_of_range &) {}                       // digital_clock_recovery_mm_ff.cc:58-59
    try { grhip_make_clock_recovery_mm_ff(2.f, -0.01f, 0.5f, 0.01f, 0.001f); fails++; }
    catch (const std::out_of_range &) {}                       // .cc:60-61
    try { grhip_make_correlate_access_code_bb(std::string(65, '1'), 0); fails++; }
    catch (const std::out_of_range &) {}                       // digital_correlate_access_code_bb.cc:54-57
    try { grhip_make_pfb_channelizer_ccf(8, std::vector<float>(16, 1.f), 3.0f); fails++; }
    catch (const std::invalid_argument &) {}                   // gr_pfb_channelizer_ccf.cc:57-60
    try { grhip_make_fft_vcc(0, true, std::vector<float>()); fails++; }
    catch (const std::out_of_range &) {}                       // gri_fft.cc:104-105
    grhip_correlate_access_code_bb_sptr c = grhip_make_correlate_access_code_bb("1011", 0);
    if (c->set_access_code(std::string(65, '0'))) fails++;      // returns false, keeps old code
    if (!c->set_access_code("110011")) fails++;
    grhip_fir_filter_ccf_sptr f = grhip_make_fir_filter_ccf(4, std::vector<float>(64, 0.5f));
    if (f->history() != 64 || f->decimation() != 4 || f->relative_rate() != 0.25) fails++;
    grhip_quadrature_demod_cf_sptr q = grhip_make_quadrature_demod_cf(1.0f);
    if (q->history() != 2) fails++;
    std::cout << "errors test: " << (fails ? "FAIL" : "ok") << "\n";
    return fails;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: host_chain_test <dir> <chain|errors|widened|tail>\n"; return 2; }
    std::string dir = argv[1], mode = argv[2];
    if (mode == "errors") return test_errors();
    if (mode == "widened") {
        // SURVEY 8f blocks through the block interface: fft_filter_ccc (output multiple nsamples) and
        // pager_slicer_fb -> unpack_k_bits_bb (gr_sync_interpolator), reference scheduler semantics for the tail
        std::vector<gr_complex> x = slurp<gr_complex>(dir + "/x.c64");
        std::vector<gr_complex> taps = slurp<gr_complex>(dir + "/taps.c64");
        std::vector<float> soft = slurp<float>(dir + "/soft.f32");
        gr_block_sptr ff = grhip_make_fft_filter_ccc(2, taps);
        grhip_linear_flowgraph g1(1 << 16, false); g1.connect(ff);
        dump(dir + "/fftfilt.c64", g1.run(x.data(), x.size()));
        grhip_pager_slicer_fb_sptr ps = grhip_make_pager_slicer_fb(0.002f);
        gr_block_sptr up = grhip_make_unpack_k_bits_bb(2);
        grhip_linear_flowgraph g2(1 << 16, false); g2.connect(ps); g2.connect(up);
        dump(dir + "/dibits.u8", g2.run(soft.data(), soft.size()));
        std::vector<float> dc(1, ps->dc_offset());
        std::vector<unsigned char> dcb((unsigned char *)dc.data(), (unsigned char *)dc.data() + 4);
        dump(dir + "/dc.f32", dcb);
        // correlator output -> framer_sink_1 -> message queue (pkt.py:143-147), 4000 items per work() call
        std::vector<unsigned char> fl = slurp<unsigned char>(dir + "/flagged.u8");
        gr_msg_queue_sptr q = gr_make_msg_queue();
        grhip_framer_sink_1_sptr fs = grhip_make_framer_sink_1(q);
        for (size_t a = 0; a < fl.size(); a += 4000) {
            gr_vector_const_void_star in(1, fl.data() + a);
            gr_vector_void_star out;
            int n = (int)std::min<size_t>(4000, fl.size() - a);
            if (fs->work(n, in, out) != n) { std::cerr << "framer work\n"; return 1; }
        }
        std::vector<unsigned char> flat;
        while (gr_message_sptr m = q->delete_head_nowait()) {
            flat.push_back((unsigned char)m->arg1());
            flat.push_back((unsigned char)(m->length() & 0xff));
            flat.push_back((unsigned char)(m->length() >> 8));
            flat.insert(flat.end(), m->msg(), m->msg() + m->length());
        }
        dump(dir + "/messages.bin", flat);
        std::cout << "widened ok\n";
        return 0;
    }

    if (mode == "tail") {
        // a SHORT finite stream under the reference's scheduler semantics (whole output multiples only, a block
        // is done once one multiple can no longer be requested; 4096-item calls = half a 64 KiB buffer):
        // with the reference's output_multiple every item comes out; the opt-in batching drops a tail
        std::vector<gr_complex> x = slurp<gr_complex>(dir + "/x.c64");
        std::vector<gr_complex> taps = slurp<gr_complex>(dir + "/taps.c64");
        std::vector<double> p = slurp<double>(dir + "/params.f64");
        {
            gr_block_sptr xl = grhip_make_freq_xlating_fir_filter_ccc((int)p[0], taps, p[1], p[2]);
            gr_block_sptr qd = grhip_make_quadrature_demod_cf((float)p[3]);
            if (xl->output_multiple() != 1 || qd->output_multiple() != 1) { std::cerr << "output_multiple\n"; return 1; }
            grhip_linear_flowgraph g(4096, false); g.connect(xl); g.connect(qd);
            dump(dir + "/demod_ref_multiple.f32", g.run(x.data(), x.size()));
        }
        {
            gr_block_sptr xl = grhip_make_freq_xlating_fir_filter_ccc((int)p[0], taps, p[1], p[2]);
            gr_block_sptr qd = grhip_make_quadrature_demod_cf((float)p[3]);
            grhip_set_batch_items(xl, 1024);
            grhip_linear_flowgraph g(4096, false); g.connect(xl); g.connect(qd);
            dump(dir + "/demod_batched.f32", g.run(x.data(), x.size()));
        }
        std::cout << "tail ok\n";
        return 0;
    }
    std::vector<gr_complex> x = slurp<gr_complex>(dir + "/x.c64");
    std::vector<gr_complex> taps = slurp<gr_complex>(dir + "/taps.c64");
    std::vector<double> p = slurp<double>(dir + "/params.f64");
    // params: decim, center_freq, fs, demod_gain, omega, gain_omega, mu, gain_mu, rel_limit, threshold
    std::vector<unsigned char> codev = slurp<unsigned char>(dir + "/code.txt");
    std::string code(codev.begin(), codev.end());

    gr_block_sptr xl = grhip_make_freq_xlating_fir_filter_ccc((int)p[0], taps, p[1], p[2]);
    gr_block_sptr qd = grhip_make_quadrature_demod_cf((float)p[3]);
    gr_block_sptr mm = grhip_make_clock_recovery_mm_ff((float)p[4], (float)p[5], (float)p[6], (float)p[7], (float)p[8]);
    gr_block_sptr sl = grhip_make_binary_slicer_fb();
    gr_block_sptr ca = grhip_make_correlate_access_code_bb(code, (int)p[9]);

    {   // stage outputs, each through its own small graph so that they can be compared one by one
        grhip_linear_flowgraph g1; g1.connect(xl); g1.connect(qd);
        std::vector<unsigned char> dem = g1.run(x.data(), x.size());
        dump(dir + "/demod.f32", dem);
        grhip_linear_flowgraph g2; g2.connect(mm);
        std::vector<unsigned char> soft = g2.run(dem.data(), dem.size() / 4);
        dump(dir + "/soft.f32", soft);
        grhip_linear_flowgraph g3; g3.connect(sl); g3.connect(ca);
        std::vector<unsigned char> bits = g3.run(soft.data(), soft.size() / 4);
        dump(dir + "/bits.u8", bits);
    }
    std::cout << "chain ok\n";
    return 0;
}
