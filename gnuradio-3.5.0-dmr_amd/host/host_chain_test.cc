// host_chain_test.cc -- C++ drop-in test: builds the DMR chain out of the grhip
// block wrappers exactly as a GNU Radio 3.5 C++ application would (factories
// returning shared pointers, connect in order, run), through the stand-in
// executor.  Reads the capture and parameters from files written by the pytest
// that drives it and writes each stage's output for comparison with the oracle.
//
// usage: host_chain_test <dir> <mode: chain|errors>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "grhip_fir_kernels.h"
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

// ---- "for each implementation in the info table" (the pattern of filter/qa_gr_fir_ccf.cc:103-177) ----------
// Every registered implementation is run on integer-valued pseudo-random data for ntaps in [0, 9] and output
// lengths in [0, 17] against a plain dot product in double; tolerance |expected| * 1e-5 as in the reference
// (qa_gr_fir_ccf.cc:54,151-152).  With the real GNU Radio the table also holds "generic" and "SSE"; here it holds
// what grhip_fir_sysconfig adds.
// the reference's generators: srandom(0), rint(uniform() * 32767) (qa_gr_fir_ccf.cc:63-85; 32768 for fff,
// qa_gr_fir_fff.cc:120-131) -- glibc's random() gives the very sequence the reference's own run uses
// (through random_r on a state of our own -- 128 bytes: the generator random() itself uses -- because random()'s state is
// process-wide and the GPU runtime's threads draw from it now and then: a run whose sequence had been shifted got other data,
// among them an output whose terms cancel to 2e-4 of their size, which no float32 sum reproduces to 1e-5 of the result)
static struct random_data qa_rd;
static char qa_rd_state[128];
static void qa_srandom(unsigned seed)
{
    memset(&qa_rd, 0, sizeof(qa_rd));
    initstate_r(seed, qa_rd_state, sizeof(qa_rd_state), &qa_rd);
}
static long qa_random() { int32_t r = 0; random_r(&qa_rd, &r); return r; }
static float qa_uniform() { return 2.0 * ((float)qa_random() / 2147483647.0 - 0.5); }
template <class T> static T rnd_item(float scale);
template <> float rnd_item<float>(float scale) { return (float)rint(qa_uniform() * scale); }
template <> gr_complex rnd_item<gr_complex>(float scale)
{
    const float re = rint(qa_uniform() * scale), im = rint(qa_uniform() * scale);
    return gr_complex(re, im);
}
template <class FIR, class INFO, class I, class O, class TAP>
static int qa_one_signature(const char *sig, void (*get_info)(std::vector<INFO> *), int MAX_TAPS, double tol, float scale)
{
    std::vector<INFO> info;
    get_info(&info);
    int fails = 0, cases = 0;
    for (auto &p : info) {
        qa_srandom(0);       // we want reproducibility (qa_gr_fir_ccf.cc:118)
        const int OUTPUT_LEN = 17, INPUT_LEN = MAX_TAPS + OUTPUT_LEN;
        for (int n = 0; n <= MAX_TAPS; n++)
            for (int ol = 0; ol <= OUTPUT_LEN; ol++) {
                std::vector<I> input(INPUT_LEN);
                std::vector<TAP> taps(MAX_TAPS);
                for (auto &v : input) v = rnd_item<I>(scale);
                for (auto &v : taps) v = rnd_item<TAP>(scale);
                std::vector<TAP> f1_taps(taps.begin(), taps.begin() + n);
                FIR *f1 = p.create(f1_taps);
                std::vector<O> actual(OUTPUT_LEN, O());
                f1->filterN(actual.data(), input.data(), ol);
                for (int o = 0; o < ol; o++) {
                    std::complex<double> sum = 0;
                    for (int i = 0; i < n; i++) sum += std::complex<double>(input[o + i]) * std::complex<double>(taps[n - i - 1]);
                    const std::complex<double> got(actual[o]);
                    if (std::abs(got - sum) > std::abs(sum) * tol) {
                        fails++;
                        std::cout << "  filterN: " << sig << " ntaps " << n << " outputs " << ol << " output " << o << ": got " << got
                                  << " expected " << sum << "\n";
                    }
                    cases++;
                }
                if (n > 0 && ol > 0) {       // filter() = one output (it may take another engine: same tolerance); get_taps()
                    const std::complex<double> one(f1->filter(input.data()));
                    std::complex<double> sum0 = 0;
                    for (int i = 0; i < n; i++) sum0 += std::complex<double>(input[i]) * std::complex<double>(taps[n - i - 1]);
                    if (std::abs(one - sum0) > std::abs(sum0) * tol) {
                        fails++;
                        std::cout << "  filter: " << sig << " ntaps " << n << " (after filterN of " << ol << "): got " << one << " expected " << sum0 << "\n";
                    }
                    if (f1->get_taps() != f1_taps || f1->ntaps() != (unsigned)n) fails++;
                }
                delete f1;
            }
        // decimating form and set_taps (filterNdec: output[i] = filter(&input[i * decimate]), gr_fir_XXX.h.t:97-100)
        {
            const int T = 37, D = 3, N = 200;
            std::vector<I> input((N - 1) * D + T);
            std::vector<TAP> t1(5), t2(T);
            for (auto &v : input) v = rnd_item<I>(scale);
            for (auto &v : t1) v = rnd_item<TAP>(scale);
            for (auto &v : t2) v = rnd_item<TAP>(scale);
            FIR *f = p.create(t1);
            f->set_taps(t2);
            std::vector<O> out(N);
            f->filterNdec(out.data(), input.data(), N, D);
            for (int o = 0; o < N; o++) {
                std::complex<double> sum = 0;
                for (int i = 0; i < T; i++) sum += std::complex<double>(input[o * D + i]) * std::complex<double>(t2[T - i - 1]);
                if (std::abs(std::complex<double>(out[o]) - sum) > std::abs(sum) * tol) {
                    fails++;
                    std::cout << "  filterNdec: " << sig << " output " << o << ": got " << std::complex<double>(out[o]) << " expected " << sum << "\n";
                }
                cases++;
            }
            delete f;
        }
        std::cout << " gr_fir_" << sig << " [" << p.name << "] " << cases << " outputs, " << fails << " wrong\n";
    }
    return fails + (info.empty() ? 1 : 0);
}

static int test_fir_qa()
{
    int fails = 0;
    // taps range and tolerance per signature as in the reference: ccf / ccc 0..9 taps, |expected| * 1e-5
    // (qa_gr_fir_ccf.cc:54,109, qa_gr_fir_ccc.cc:54); fff 0..32 taps, |expected| * 9e-3 (qa_gr_fir_fff.cc:147,189-190)
    fails += qa_one_signature<gr_fir_ccf, gr_fir_ccf_info, gr_complex, gr_complex, float>("ccf", grhip_fir_sysconfig::get_gr_fir_ccf_info, 9, 1e-5, 32767.f);
    fails += qa_one_signature<gr_fir_fff, gr_fir_fff_info, float, float, float>("fff", grhip_fir_sysconfig::get_gr_fir_fff_info, 32, 9e-3, 32768.f);
    fails += qa_one_signature<gr_fir_ccc, gr_fir_ccc_info, gr_complex, gr_complex, gr_complex>("ccc", grhip_fir_sysconfig::get_gr_fir_ccc_info, 9, 1e-5, 32767.f);
    std::cout << "fir qa: " << (fails ? "FAIL" : "ok") << "\n";
    return fails;
}

// N-port adapters and the FFT block on its abstract base, through work() as the scheduler calls it
static int test_adapters()
{
    int fails = 0;
    const int NS = 5, N = 1000;
    std::vector<gr_complex> x(NS * N);
    for (int i = 0; i < NS * N; ++i) x[i] = gr_complex((float)i, (float)-i);
    auto s2s = grhip_make_adapter<grhip_stream_to_streams_blk>(sizeof(gr_complex), (size_t)NS);
    if (s2s->decimation() != (unsigned)NS || s2s->output_signature()->max_streams() != NS) fails++;
    std::vector<std::vector<gr_complex>> streams(NS, std::vector<gr_complex>(N));
    {
        gr_vector_const_void_star in(1, x.data());
        gr_vector_void_star out(NS);
        for (int j = 0; j < NS; ++j) out[j] = streams[j].data();
        if (s2s->work(N, in, out) != N) fails++;
        for (int j = 0; j < NS; ++j)
            for (int i = 0; i < N; ++i)
                if (streams[j][i] != x[i * NS + j]) { fails++; break; }
    }
    auto back = grhip_make_adapter<grhip_streams_to_stream_blk>(sizeof(gr_complex), (size_t)NS);
    {
        std::vector<gr_complex> y(NS * N);
        gr_vector_const_void_star in(NS);
        for (int j = 0; j < NS; ++j) in[j] = streams[j].data();
        gr_vector_void_star out(1, y.data());
        if (back->interpolation() != (unsigned)NS || back->work(NS * N, in, out) != NS * N || y != x) fails++;
    }
    auto v2s = grhip_make_adapter<grhip_vector_to_streams_blk>(sizeof(gr_complex), (size_t)NS);
    {
        std::vector<std::vector<gr_complex>> o(NS, std::vector<gr_complex>(N));
        gr_vector_const_void_star in(1, x.data());
        gr_vector_void_star out(NS);
        for (int j = 0; j < NS; ++j) out[j] = o[j].data();
        if (v2s->work(N, in, out) != N || o != streams) fails++;
    }
    auto s2v = grhip_make_adapter<grhip_stream_to_vector_blk>(sizeof(gr_complex), (size_t)100);
    {
        std::vector<gr_complex> y(NS * N);
        gr_vector_const_void_star in(1, x.data());
        gr_vector_void_star out(1, y.data());
        if (s2v->decimation() != 100 || s2v->work(NS * N / 100, in, out) != NS * N / 100 || y != x) fails++;
    }
    auto hd = grhip_make_adapter<grhip_head_blk>(sizeof(gr_complex), 1500ull);
    {
        std::vector<gr_complex> y(NS * N);
        gr_vector_const_void_star in(1, x.data());
        gr_vector_void_star out(1, y.data());
        if (hd->work(1000, in, out) != 1000) fails++;
        gr_vector_const_void_star in2(1, x.data() + 1000);
        gr_vector_void_star out2(1, y.data() + 1000);
        if (hd->work(1000, in2, out2) != 500) fails++;
        if (hd->work(1000, in2, out2) != -1) fails++;                      // WORK_DONE, as gr_head.cc:49-50
        for (int i = 0; i < 1500; ++i) if (y[i] != x[i]) { fails++; break; }
        hd->reset();
        if (hd->work(10, in, out) != 10) fails++;
    }
    // gr_fft_vcc_hip on the abstract base: the base class's (non-virtual) set_window reaches the device
    {
        const int F = 64;
        gr_fft_vcc_hip_sptr f = gr_make_fft_vcc_hip(F, true, std::vector<float>());
        gr_fft_vcc *base = f.get();
        std::vector<gr_complex> in1(F, gr_complex(1.f, 0.f)), o1(F), o2(F);
        gr_vector_const_void_star in(1, in1.data());
        gr_vector_void_star out(1, o1.data());
        if (f->work(1, in, out) != 1 || std::abs(o1[0] - gr_complex((float)F, 0.f)) > 1e-3f) fails++;
        if (base->set_window(std::vector<float>(3, 1.f))) fails++;          // wrong size: refused (gr_fft_vcc.cc:57-63)
        if (!base->set_window(std::vector<float>(F, 0.5f))) fails++;
        gr_vector_void_star out2(1, o2.data());
        if (f->work(1, in, out2) != 1 || std::abs(o2[0] - gr_complex(0.5f * F, 0.f)) > 1e-3f) fails++;
        try { gr_make_fft_vcc_hip(0, true, std::vector<float>()); fails++; } catch (const std::out_of_range &) {}
    }
    std::cout << "adapters: " << (fails ? "FAIL" : "ok") << "\n";
    return fails;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: host_chain_test <dir> <chain|errors|widened|tail|firqa|adapters>\n"; return 2; }
    std::string dir = argv[1], mode = argv[2];
    if (mode == "errors") return test_errors();
    if (mode == "firqa") return test_fir_qa();
    if (mode == "adapters") return test_adapters();
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
