// grhip_executor.h -- a single-threaded stand-in for what the GNU Radio runtime
// does around a linear chain of blocks, enough to drive the wrappers exactly the
// way the scheduler would:
//   * every edge is a buffer; each reader starts with history-1 zeros in front
//     (runtime/gr_flat_flowgraph.cc:150, runtime/gr_buffer.cc:200-213)
//   * a block is offered noutput_items (a multiple of output_multiple, capped by
//     `max_noutput`), forecast() says how many inputs that needs, the request is
//     halved until it fits what is available (runtime/gr_block_executor.cc:302-348)
//   * general_work() -> produce n, consume consumed() (consume_each)
//   * a block that returns 0 is simply called again (parameter updates)
// Not a scheduler: no threads, one input and one output per block except for the
// source (none) and sink; the multi-input PFB is driven directly in its own test.
#pragma once
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "gr_shim.h"

class grhip_linear_flowgraph {
    struct edge { std::vector<unsigned char> data; size_t item = 1; size_t rd = 0; };   // rd in items
    std::vector<gr_block_sptr> d_blocks;
    int d_max_noutput;
    bool d_drain_tail;
public:
    // drain_tail = true: once the source is exhausted, what is left in front of a block is pushed through
    // item by item so that every output of a finite test vector is produced.  false: the reference's
    // behaviour -- a block whose forecast for one output_multiple cannot be met any more is done
    // (runtime/gr_block_executor.cc:335-348); needed for blocks that insist on whole output multiples.
    explicit grhip_linear_flowgraph(int max_noutput = 1 << 20, bool drain_tail = true)
        : d_max_noutput(max_noutput), d_drain_tail(drain_tail) {}
    void connect(gr_block_sptr b) { d_blocks.push_back(b); }

    // runs `input` (n_in items of the first block's input size) through the chain,
    // returns the last block's output bytes
    std::vector<unsigned char> run(const void *input, size_t n_in)
    {
        std::vector<edge> e(d_blocks.size() + 1);
        for (size_t i = 0; i < d_blocks.size(); ++i) {
            e[i].item = d_blocks[i]->input_signature()->sizeof_stream_item(0);
            size_t hist = d_blocks[i]->history() - 1;
            e[i].data.assign(hist * e[i].item, 0);            // history zeros
        }
        e.back().item = d_blocks.back()->output_signature()->sizeof_stream_item(0);
        e[0].data.insert(e[0].data.end(), (const unsigned char *)input,
                         (const unsigned char *)input + n_in * e[0].item);
        std::vector<bool> upstream_done(d_blocks.size() + 1, false);
        upstream_done[0] = true;
        bool progress = true;
        while (progress) {
            progress = false;
            for (size_t i = 0; i < d_blocks.size(); ++i) {
                gr_block &b = *d_blocks[i];
                edge &in = e[i], &out = e[i + 1];
                for (int guard = 0; guard < 1000000; ++guard) {
                    size_t avail = in.data.size() / in.item - in.rd;
                    int mult = b.output_multiple();
                    int nout = (d_max_noutput / mult) * mult;
                    if (nout < mult) nout = mult;
                    gr_vector_int req(1);
                    // shrink the request until its forecast fits (gr_block_executor.cc:313-348)
                    while (true) {
                        b.forecast(nout, req);
                        if ((size_t)req[0] <= avail) break;
                        if (nout <= mult) {
                            // below one output_multiple: allowed only when upstream is finished
                            if (!upstream_done[i] || mult == 1 || !d_drain_tail) { nout = 0; break; }
                            mult = 1;                      // drain the tail item by item
                            nout = std::max(1, nout / 2);
                            continue;
                        }
                        nout = ((nout / 2) / mult) * mult;
                        if (nout < mult) nout = mult;
                    }
                    if (nout <= 0) break;
                    b.forecast(nout, req);
                    if ((size_t)req[0] > avail) break;
                    size_t old = out.data.size();
                    out.data.resize(old + (size_t)nout * out.item);
                    gr_vector_int ninput(1, (int)avail);
                    gr_vector_const_void_star ins(1, in.data.data() + in.rd * in.item);
                    gr_vector_void_star outs(1, out.data.data() + old);
                    int hist_before = (int)b.history();
                    int n = b.general_work(nout, ninput, ins, outs);
                    if (n < 0) throw std::runtime_error("block returned " + std::to_string(n));
                    out.data.resize(old + (size_t)n * out.item);
                    in.rd += b.consumed();
                    if ((int)b.history() != hist_before) {
                        // the block changed its history (new taps): the scheduler keeps the read
                        // pointer so that the newest consumed item stays aligned
                        long delta = (long)b.history() - hist_before;
                        if (delta > 0 && in.rd < (size_t)delta) {
                            in.data.insert(in.data.begin(), ((size_t)delta - in.rd) * in.item, 0);
                            in.rd = 0;
                        } else {
                            in.rd -= delta;
                        }
                    }
                    if (n > 0 || b.consumed() > 0) progress = true;
                    else if ((int)b.history() == hist_before && n == 0 && guard > 2) break;
                }
                upstream_done[i + 1] = upstream_done[i];
            }
        }
        return e.back().data;
    }
};
