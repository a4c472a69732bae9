// capi_framer.hip -- gr_framer_sink_1: kernels and C ABI.
//
// Reference: gnuradio-core/src/lib/general/gr_framer_sink_1.{h,cc}.  The block is a bit-serial state
// machine (search for the correlator's flag bit .cc:105-117; shift 32 header bits, the flagged bit first
// .cc:119-152; shift 8*len payload bits into bytes .cc:154-182), but its state only changes per PACKET:
//   1. framer_pack_kernel     every input byte once: bit 0 and bit 1 of 32 items packed MSB-first into one
//                             data word and one flag word (1 B read, 1/4 B written per item);
//   2. framer_walk_kernel     one wavefront per stream walks packet to packet: 64 flag/data words (2048
//                             items) live in the lanes' registers, the next flag is a ballot + clz, the
//                             header a funnel shift of two data words; it emits message records and
//                             payload jobs and carries the reference's state (partial header, partial
//                             payload) across calls in device memory;
//   3. framer_payload_kernel  the payload bytes of all jobs, one lane per byte, from the packed data words.
// Long calls replace step 2 by the segment-parallel walk further down (framer_segwalk / fixup / emit kernels).
// Messages (whitener offset = gr_message arg1, payload) collect in a device pool until the host fetches them.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

#include "grhip_internal.h"

using namespace grhip;

namespace grhip {

enum { FR_SEARCH = 0, FR_HAVE_SYNC = 1, FR_HAVE_HEADER = 2 };   // gr_framer_sink_1.h:62

struct FramerState {
    int mode;
    unsigned header;        // d_header
    int hdr_cnt;            // d_headerbitlen_cnt
    int pktlen;             // d_packetlen (bytes)
    int woff;               // d_packet_whitener_offset
    int bits_done;          // 8 * d_packetlen_cnt + d_packet_byte_index
    unsigned msg_count;     // complete messages in msgs[]
    unsigned pool_used;     // bytes of pool in use (including the open packet's reservation)
    unsigned open_off;      // pool offset of the packet being filled
    unsigned njobs;         // payload jobs of the current call
    unsigned pad[6];
};
struct FramerMsg { unsigned woff, len, off, pad; };
struct FramerJob { unsigned src_bit, nbits, dst_off, dst_bit; };

// 4 items (one per byte, first item in the low byte) -> 4 bits, first item most significant
__device__ inline unsigned nib(unsigned w, int bit)
{
    return ((((w >> bit) & 0x01010101u) * 0x08040201u) >> 24) & 0xfu;
}

__device__ __forceinline__ void framer_pack_body(const unsigned char *__restrict__ in, long long n, unsigned *__restrict__ F,
                                                 unsigned *__restrict__ D, long long nwords_padded)
{
    for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < nwords_padded; w += (long long)gridDim.x * 256) {
        const long long b = w << 5;
        unsigned f = 0, d = 0;
        if (b + 32 <= n && ((((uintptr_t)(in + b)) & 15) == 0)) {
            const uint4 q0 = *reinterpret_cast<const uint4 *>(in + b), q1 = *reinterpret_cast<const uint4 *>(in + b + 16);
            const unsigned v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                d = (d << 4) | nib(v[i], 0);
                f = (f << 4) | nib(v[i], 1);
            }
        } else {
            for (int i = 0; i < 32; ++i) {
                const unsigned c = b + i < n ? in[b + i] : 0u;
                d = (d << 1) | (c & 1u);
                f = (f << 1) | ((c >> 1) & 1u);
            }
        }
        F[w] = f;
        D[w] = d;
    }
}

__global__ void __launch_bounds__(256)
framer_pack_kernel(const unsigned char *__restrict__ in, long long n, unsigned *__restrict__ F, unsigned *__restrict__ D,
                   long long nwords_padded)
{
    framer_pack_body(in, n, F, D, nwords_padded);
}

// multi-capture form: blockIdx.y = stream; stream s has min(n_max, n_ptr[s * n_stride]) items (n_ptr may be null)
__device__ __forceinline__ long long batch_items(const int *n_ptr, int n_stride, long long n_max, int s)
{
    if (!n_ptr) return n_max;
    const long long v = n_ptr[(long long)s * n_stride];
    return v < 0 ? 0 : (v < n_max ? v : n_max);
}
__global__ void __launch_bounds__(256)
framer_pack_batch_kernel(const unsigned char *__restrict__ in, long long in_stride, const int *n_ptr, int n_stride, long long n_max,
                         unsigned *__restrict__ F, unsigned *__restrict__ D, long long words_stride)
{
    const int s = blockIdx.y;
    framer_pack_body(in + (long long)s * in_stride, batch_items(n_ptr, n_stride, n_max, s), F + (long long)s * words_stride,
                     D + (long long)s * words_stride, words_stride);
}

__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// One wavefront: the reference's state machine at packet granularity.  Every branch is wave-uniform and the
// bookkeeping is 32-bit scalar arithmetic (a lone wave issues about one instruction every five cycles, so
// the instruction count per packet is what the walk costs).  F and D hold nwords = ceil(n/32) words plus two
// zero words.  n < 2^31.
__device__ __forceinline__ void framer_walk_body(const unsigned *__restrict__ F, const unsigned *__restrict__ D, unsigned n,
                                                 FramerState *S, FramerMsg *msgs, FramerJob *jobs)
{
    const unsigned lane = threadIdx.x;
    int mode = uni(S->mode), hdr_cnt = uni(S->hdr_cnt), pktlen = uni(S->pktlen), woff = uni(S->woff), bits_done = uni(S->bits_done);
    unsigned header = uni(S->header), msg_count = uni(S->msg_count), pool_used = uni(S->pool_used), open_off = uni(S->open_off);
    unsigned njobs = 0;
    const unsigned nwords = (n + 31u) >> 5;
    unsigned pos = 0, wb = 0;                    // wb: first word of the register window
    bool have_window = false;
    unsigned Fw = 0, Dw = 0;

    auto load_window = [&](unsigned w0) {
        wb = w0;
        have_window = true;
        const unsigned wi = wb + lane;
        Fw = wi < nwords ? F[wi] : 0u;
        Dw = wi < nwords ? D[wi] : 0u;
    };
    // k (1..32) data bits starting at item s, first item most significant
    auto bits_at = [&](unsigned s, int k) -> unsigned {
        const unsigned i = s >> 5;
        const int o = (int)(s & 31u);
        unsigned d0, d1;
        if (have_window && i >= wb && i + 1u < wb + 64u) {
            const int li = (int)(i - wb);
            d0 = __builtin_amdgcn_readlane(Dw, li);
            d1 = __builtin_amdgcn_readlane(Dw, li + 1);
        } else {
            d0 = D[i]; d1 = D[i + 1];            // zero padded behind nwords
        }
        const unsigned long long w = ((unsigned long long)d0 << 32) | d1;
        return (unsigned)((w << o) >> (64 - k));
    };

    for (;;) {
        if (mode == FR_HAVE_SYNC) {
            const unsigned avail = n - pos;
            const unsigned need = 32u - (unsigned)hdr_cnt;
            const int take = (int)(avail < need ? avail : need);
            if (take > 0) {
                const unsigned b = bits_at(pos, take);
                header = take == 32 ? b : ((header << take) | b);
                hdr_cnt += take;
                pos += (unsigned)take;
            }
            if (hdr_cnt < 32) break;
            if (((header >> 16) ^ (header & 0xffffu)) == 0) {          // header_ok(), .h:85-89
                pktlen = (int)((header >> 16) & 0x0fffu);                // header_payload(), .h:91-98
                woff = (int)((header >> 28) & 0xfu);
                if (pktlen == 0) {                                       // .cc:137-146
                    if (lane == 0) msgs[msg_count] = FramerMsg{(unsigned)woff, 0u, pool_used, 0u};
                    ++msg_count;
                    mode = FR_SEARCH;
                } else {
                    mode = FR_HAVE_HEADER;
                    bits_done = 0;
                    open_off = pool_used;
                    pool_used += (unsigned)pktlen;
                }
            } else {
                mode = FR_SEARCH;                                        // bad header, .cc:148-149
            }
        } else if (mode == FR_HAVE_HEADER) {
            const unsigned avail = n - pos;
            const unsigned rem = (unsigned)(8 * pktlen - bits_done);
            const unsigned take = avail < rem ? avail : rem;
            if (take > 0) {
                if (lane == 0) jobs[njobs] = FramerJob{pos, take, open_off, (unsigned)bits_done};
                ++njobs;
                bits_done += (int)take;
                pos += take;
            }
            if (bits_done < 8 * pktlen) break;
            if (lane == 0) msgs[msg_count] = FramerMsg{(unsigned)woff, (unsigned)pktlen, open_off, 0u};
            ++msg_count;
            mode = FR_SEARCH;
        } else {
            // next flagged item at or after pos (.cc:109-116: the flagged item is the first header bit)
            bool found = false;
            while (pos < n) {
                const unsigned w0 = pos >> 5;
                if (!have_window || w0 < wb || w0 >= wb + 64u) load_window(w0);
                const unsigned wi = wb + lane;
                unsigned f = Fw;
                if (wi < w0) f = 0u;
                else if (wi == w0) f &= 0xffffffffu >> (pos & 31u);
                const unsigned long long bal = __ballot(f != 0u);
                if (bal == 0ull) { pos = (wb + 64u) << 5; continue; }
                const int L = __ffsll((long long)bal) - 1;
                const unsigned word = __builtin_amdgcn_readlane(f, L);
                pos = ((wb + (unsigned)L) << 5) + (unsigned)__clz((int)word);
                found = true;
                break;
            }
            if (!found) break;                                           // pos >= n: still searching
            // fast path: header and payload both inside this call -> one pass, no state round trip
            if (pos + 32u <= n) {
                const unsigned h = bits_at(pos, 32);
                const unsigned len = (h >> 16) & 0x0fffu;
                if (((h >> 16) ^ (h & 0xffffu)) != 0) { pos += 32u; continue; }
                if (pos + 32u + 8u * len <= n) {
                    if (lane == 0) {
                        msgs[msg_count] = FramerMsg{(h >> 28) & 0xfu, len, pool_used, 0u};
                        if (len) jobs[njobs] = FramerJob{pos + 32u, 8u * len, pool_used, 0u};
                    }
                    ++msg_count;
                    if (len) ++njobs;
                    pool_used += len;
                    pos += 32u + 8u * len;
                    continue;
                }
            }
            mode = FR_HAVE_SYNC;                                         // enter_have_sync(), .cc:46-55
            header = 0u;
            hdr_cnt = 0;
        }
    }
    if (lane == 0) {
        S->mode = mode; S->header = header; S->hdr_cnt = hdr_cnt; S->pktlen = pktlen; S->woff = woff;
        S->bits_done = bits_done; S->msg_count = msg_count; S->pool_used = pool_used; S->open_off = open_off;
        S->njobs = njobs;
    }
}

__global__ void __launch_bounds__(64)
framer_walk_kernel(const unsigned *__restrict__ F, const unsigned *__restrict__ D, unsigned n, FramerState *S,
                   FramerMsg *msgs, FramerJob *jobs)
{
    framer_walk_body(F, D, n, S, msgs, jobs);
}

// multi-capture form: one wavefront per stream (blockIdx.x = stream), every stream from the search state
__global__ void __launch_bounds__(64)
framer_walk_batch_kernel(const unsigned *__restrict__ F, const unsigned *__restrict__ D, long long words_stride, const int *n_ptr,
                         int n_stride, long long n_max, FramerState *S, FramerMsg *msgs, FramerJob *jobs, long long rec_stride)
{
    const int s = blockIdx.x;
    FramerState *st = S + s;
    if (threadIdx.x == 0) *st = FramerState{};          // enter_search(): a capture is framed whole, from a fresh block
    __syncthreads();
    framer_walk_body(F + (long long)s * words_stride, D + (long long)s * words_stride, (unsigned)batch_items(n_ptr, n_stride, n_max, s), st,
                     msgs + (long long)s * rec_stride, jobs + (long long)s * rec_stride);
}

// ---- segment-parallel walk (long calls) --------------------------------------------------------------
// The walk is serial only through "where does the search resume".  Every segment of `seg` items is walked by its
// own wavefront as if the search state held at its first item (framer_segwalk_kernel: one record per packet,
// with running message / payload-byte counts).  A single wavefront then goes through the segments in order
// (framer_fixup_kernel): when the true resume position lies at or before a segment's start, or in one of the
// gaps between the packets the speculative walk found, both walks see the same next flag and everything from
// there on is taken over wholesale; only when it falls inside a speculative packet does the fix-up walk packets
// itself until the two coincide.  framer_emit_kernel finally turns the accepted records into message records
// and payload jobs at the indices the fix-up assigned.
struct FrRec { unsigned p, h, cmsg, cbytes; };   // flag position, header bits, messages / payload bytes before it in the segment
struct FrSeg { unsigned cnt, totm, totb, exit, acc, mbase, bbase, lastp, lasth, lastcm, lastcb, firstp; };   // last*: the segment's last record; firstp: its first flag

__device__ inline bool fr_good(unsigned h) { return ((h >> 16) ^ (h & 0xffffu)) == 0; }
__device__ inline unsigned fr_len(unsigned h) { return (h >> 16) & 0x0fffu; }

// next flagged item in [pos, lim), or lim; F zero padded behind n
__device__ inline unsigned fr_next_flag(const unsigned *__restrict__ F, unsigned nwords, unsigned pos, unsigned lim, unsigned lane)
{
    while (pos < lim) {
        const unsigned w0 = pos >> 5, wi = w0 + lane;
        unsigned f = wi < nwords ? F[wi] : 0u;
        if (lane == 0) f &= 0xffffffffu >> (pos & 31u);
        const unsigned long long bal = __ballot(f != 0u);
        if (bal == 0ull) { pos = (w0 + 64u) << 5; continue; }
        const int L = __ffsll((long long)bal) - 1;
        const unsigned word = __builtin_amdgcn_readlane(f, L);
        const unsigned p = ((w0 + (unsigned)L) << 5) + (unsigned)__clz((int)word);
        return p < lim ? p : lim;
    }
    return lim;
}

// k (1..32) data bits starting at item s (wave-uniform address: scalar loads)
__device__ inline unsigned fr_bits(const unsigned *__restrict__ D, unsigned s, int k)
{
    const unsigned i = s >> 5;
    const unsigned long long w = ((unsigned long long)D[i] << 32) | D[i + 1];
    return (unsigned)((w << (s & 31u)) >> (64 - k));
}

__global__ void __launch_bounds__(64)
framer_segwalk_kernel(const unsigned *__restrict__ F, const unsigned *__restrict__ D, unsigned n, unsigned seg,
                      unsigned reccap, FrRec *__restrict__ recs, FrSeg *__restrict__ segs)
{
    const unsigned lane = threadIdx.x, k = blockIdx.x;
    const unsigned nwords = (n + 31u) >> 5;
    const unsigned s0 = k * seg, s1 = min(s0 + seg, n);
    FrRec *r = recs + (size_t)k * reccap;
    unsigned pos = s0, cnt = 0, cm = 0, cb = 0, lp = 0, lh = 0, lcm = 0, lcb = 0, fp = 0;
    while (pos < s1) {
        const unsigned p = fr_next_flag(F, nwords, pos, s1, lane);
        if (p >= s1) { pos = s1; break; }
        if (p + 32u > n) {                                  // header cut off by the end of the call
            const unsigned hb = fr_bits(D, p, (int)(n - p));
            if (lane == 0) r[cnt] = FrRec{p, hb, cm, cb};
            if (cnt == 0) fp = p;
            lp = p; lh = hb; lcm = cm; lcb = cb;
            ++cnt;
            pos = p + 32u;
            break;
        }
        const unsigned h = fr_bits(D, p, 32);
        if (lane == 0) r[cnt] = FrRec{p, h, cm, cb};
        if (cnt == 0) fp = p;
        lp = p; lh = h; lcm = cm; lcb = cb;
        ++cnt;
        pos = p + 32u;
        if (fr_good(h)) { ++cm; cb += fr_len(h); pos += 8u * fr_len(h); }
    }
    if (lane == 0) segs[k] = FrSeg{cnt, cm, cb, pos, 0u, 0u, 0u, lp, lh, lcm, lcb, fp};
}

// where the search resumes after the packet of record (p, h); a header cut off by the end of the call ends behind it
__device__ inline unsigned fr_end(unsigned p, unsigned h, unsigned n)
{
    if (p + 32u > n || !fr_good(h)) return p + 32u;
    return p + 32u + 8u * fr_len(h);
}

__global__ void __launch_bounds__(64)
framer_fixup_kernel(const unsigned *__restrict__ F, const unsigned *__restrict__ D, unsigned n, unsigned seg, unsigned nseg,
                    unsigned reccap, const FrRec *__restrict__ recs, FrSeg *__restrict__ segs, FramerState *S,
                    FramerMsg *msgs, FramerJob *jobs)
{
    const unsigned lane = threadIdx.x;
    const unsigned nwords = (n + 31u) >> 5;
    int mode = uni(S->mode), hdr_cnt = uni(S->hdr_cnt), pktlen = uni(S->pktlen), woff = uni(S->woff), bits_done = uni(S->bits_done);
    unsigned header = uni(S->header), M = uni(S->msg_count), B = uni(S->pool_used), open_off = uni(S->open_off);
    unsigned pos = 0;
    if (lane == 0) jobs[0] = FramerJob{0u, 0u, 0u, 0u};      // slot of a packet continued from the previous call

    // ---- the packet the previous call left open (same steps as framer_walk_kernel) ----
    bool finished = false;                                   // the call ends inside that packet
    if (mode == FR_HAVE_SYNC) {
        const unsigned need = 32u - (unsigned)hdr_cnt;
        const int take = (int)(n < need ? n : need);
        const unsigned b = fr_bits(D, 0u, take);
        header = take == 32 ? b : ((header << take) | b);
        hdr_cnt += take;
        pos = (unsigned)take;
        if (hdr_cnt < 32) finished = true;
        else if (fr_good(header)) {
            pktlen = (int)fr_len(header);
            woff = (int)((header >> 28) & 0xfu);
            if (pktlen == 0) {
                if (lane == 0) msgs[M] = FramerMsg{(unsigned)woff, 0u, B, 0u};
                ++M;
                mode = FR_SEARCH;
            } else {
                mode = FR_HAVE_HEADER; bits_done = 0; open_off = B; B += (unsigned)pktlen;
            }
        } else mode = FR_SEARCH;
    }
    if (!finished && mode == FR_HAVE_HEADER) {
        const unsigned avail = n - pos, rem = (unsigned)(8 * pktlen - bits_done);
        const unsigned take = avail < rem ? avail : rem;
        if (lane == 0) jobs[0] = FramerJob{pos, take, open_off, (unsigned)bits_done};
        bits_done += (int)take;
        pos += take;
        if (bits_done < 8 * pktlen) finished = true;
        else {
            if (lane == 0) msgs[M] = FramerMsg{(unsigned)woff, (unsigned)pktlen, open_off, 0u};
            ++M;
            mode = FR_SEARCH;
        }
    }
    const unsigned M0 = M;                                   // jobs[1 + (message index - M0)] belong to this call's packets
    // the last packet of the call, to derive the state it leaves: (p, h, message index, pool offset)
    bool have_last = false;
    unsigned lp = 0, lh = 0, lm = 0, lo = 0;

    if (!finished) {
        for (unsigned kb = 0; kb < nseg; kb += 64u) {
            // the descriptors of 64 segments, one per lane: the common case below needs nothing else
            FrSeg mine = FrSeg{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            if (kb + lane < nseg) mine = segs[kb + lane];
            unsigned my_acc = 0, my_mb = 0, my_bb = 0;
            const unsigned kend = min(64u, nseg - kb);
            for (unsigned kk = 0; kk < kend; ++kk) {
                const unsigned k = kb + kk;
                const int kl = (int)kk;
                const unsigned c = __builtin_amdgcn_readlane(mine.cnt, kl), s1 = min((k + 1u) * seg, n);
                unsigned acc = c, mb = 0, bb = 0;            // first accepted record (c = none)
                if (c != 0u && pos <= __builtin_amdgcn_readlane(mine.firstp, kl)) {
                    // the search resumes at or before the segment's first flag (no flag lies between the segment's
                    // start and it): both walks take that flag next, the whole segment is taken over
                    acc = 0; mb = M; bb = B;
                    have_last = true;
                    lp = __builtin_amdgcn_readlane(mine.lastp, kl); lh = __builtin_amdgcn_readlane(mine.lasth, kl);
                    lm = M + __builtin_amdgcn_readlane(mine.lastcm, kl); lo = B + __builtin_amdgcn_readlane(mine.lastcb, kl);
                    M += __builtin_amdgcn_readlane(mine.totm, kl);
                    B += __builtin_amdgcn_readlane(mine.totb, kl);
                    pos = __builtin_amdgcn_readlane(mine.exit, kl);
                } else if (c != 0u && pos < s1) {
                    const FrRec *r = recs + (size_t)k * reccap;
                    for (;;) {
                        // first record at or after pos
                        unsigned i = c;
                        for (unsigned b = 0; b < c; b += 64u) {
                            const unsigned idx = b + lane;
                            const unsigned pv = idx < c ? r[idx].p : 0xffffffffu;
                            const unsigned long long bal = __ballot(pv >= pos);
                            if (bal) { i = b + (unsigned)(__ffsll((long long)bal) - 1); break; }
                        }
                        bool diverged = false;
                        if (i > 0) {
                            const FrRec q = r[i - 1];
                            diverged = fr_end(uni(q.p), uni(q.h), n) > pos;      // pos inside a speculative packet
                        }
                        if (!diverged) { acc = i; break; }
                        // walk one packet from the true position
                        const unsigned p = fr_next_flag(F, nwords, pos, s1, lane);
                        if (p >= s1) { acc = c; break; }     // nothing more starts in this segment
                        unsigned h;
                        if (p + 32u > n) h = fr_bits(D, p, (int)(n - p));
                        else h = fr_bits(D, p, 32);
                        have_last = true; lp = p; lh = h; lm = M; lo = B;
                        if (p + 32u <= n && fr_good(h)) {
                            const unsigned len = fr_len(h), room = n - (p + 32u);
                            if (lane == 0) {
                                msgs[M] = FramerMsg{(h >> 28) & 0xfu, len, B, 0u};
                                jobs[1u + (M - M0)] = FramerJob{p + 32u, min(8u * len, room), B, 0u};
                            }
                            ++M; B += len;
                        }
                        pos = fr_end(p, h, n);
                        if (pos >= s1) { acc = c; break; }
                    }
                    if (acc < c) {
                        const FrRec a = r[acc];
                        mb = M - uni(a.cmsg); bb = B - uni(a.cbytes);
                        have_last = true;
                        lp = __builtin_amdgcn_readlane(mine.lastp, kl); lh = __builtin_amdgcn_readlane(mine.lasth, kl);
                        lm = mb + __builtin_amdgcn_readlane(mine.lastcm, kl); lo = bb + __builtin_amdgcn_readlane(mine.lastcb, kl);
                        M += __builtin_amdgcn_readlane(mine.totm, kl) - uni(a.cmsg);
                        B += __builtin_amdgcn_readlane(mine.totb, kl) - uni(a.cbytes);
                        pos = __builtin_amdgcn_readlane(mine.exit, kl);
                    }
                }
                if (lane == kk) { my_acc = acc; my_mb = mb; my_bb = bb; }
            }
            if (kb + lane < nseg) { segs[kb + lane].acc = my_acc; segs[kb + lane].mbase = my_mb; segs[kb + lane].bbase = my_bb; }
        }
        // the state the last packet leaves (.cc:119-182)
        mode = FR_SEARCH;
        if (have_last && fr_end(lp, lh, n) > n) {
            if (lp + 32u > n) {                              // header incomplete
                mode = FR_HAVE_SYNC; header = lh; hdr_cnt = (int)(n - lp);
            } else {                                         // good header, payload incomplete: the message stays open
                mode = FR_HAVE_HEADER; pktlen = (int)fr_len(lh); woff = (int)((lh >> 28) & 0xfu);
                bits_done = (int)(n - (lp + 32u)); open_off = lo;
                M = lm;                                      // it is the last one: not complete yet
            }
        }
    }
    if (finished)                                            // nothing of this call's segments is used
        for (unsigned k = lane; k < nseg; k += 64u) segs[k].acc = segs[k].cnt;
    if (lane == 0) {
        S->mode = mode; S->header = header; S->hdr_cnt = hdr_cnt; S->pktlen = pktlen; S->woff = woff;
        S->bits_done = bits_done; S->msg_count = M; S->pool_used = B; S->open_off = open_off;
        // every message of this call has its job slot; an open last packet has one more
        S->njobs = finished ? 1u : (1u + (M - M0) + (mode == FR_HAVE_HEADER ? 1u : 0u));
        S->pad[0] = M0;
    }
}

__global__ void __launch_bounds__(256)
framer_emit_kernel(unsigned n, unsigned reccap, const FrRec *__restrict__ recs, const FrSeg *__restrict__ segs,
                   const FramerState *S, FramerMsg *msgs, FramerJob *jobs)
{
    const unsigned k = blockIdx.x;
    const FrSeg sg = segs[k];
    const unsigned M0 = S->pad[0];
    const FrRec *r = recs + (size_t)k * reccap;
    for (unsigned j = sg.acc + threadIdx.x; j < sg.cnt; j += 256u) {
        const FrRec q = r[j];
        if (q.p + 32u > n || !fr_good(q.h)) continue;
        const unsigned len = fr_len(q.h), idx = sg.mbase + q.cmsg, off = sg.bbase + q.cbytes;
        msgs[idx] = FramerMsg{(q.h >> 28) & 0xfu, len, off, 0u};
        jobs[1u + (idx - M0)] = FramerJob{q.p + 32u, min(8u * len, n - (q.p + 32u)), off, 0u};
    }
}

// payload bits -> bytes (.cc:158-162): one lane per packet byte; a byte shared with the previous call's
// job keeps the bits that are already there
__device__ __forceinline__ void framer_payload_body(const unsigned *__restrict__ D, const FramerState *S,
                                                    const FramerJob *__restrict__ jobs, unsigned char *__restrict__ pool)
{
    const unsigned njobs = S->njobs;
    for (unsigned j = blockIdx.x; j < njobs; j += gridDim.x) {
        const FramerJob jb = jobs[j];
        if (jb.nbits == 0) continue;                                     // (zero-length packet / nothing of it in this call)
        const unsigned first = jb.dst_bit >> 3, last = (jb.dst_bit + jb.nbits - 1) >> 3;
        for (unsigned y = first + threadIdx.x; y <= last; y += 256) {
            const unsigned b0 = max(8u * y, jb.dst_bit), b1 = min(8u * y + 8u, jb.dst_bit + jb.nbits);
            const unsigned k = b1 - b0;                                  // 1..8 bits of this byte
            const unsigned long long s = (unsigned long long)jb.src_bit + (b0 - jb.dst_bit);
            const unsigned long long i = s >> 5;
            const unsigned long long w = ((unsigned long long)D[i] << 32) | D[i + 1];
            const unsigned v = (unsigned)((w << (s & 31)) >> (64 - k));
            const unsigned sh = 8u * y + 8u - b1;                        // free low bits of the byte
            unsigned char *dst = pool + jb.dst_off + y;
            unsigned out = v << sh;
            if (b0 > 8u * y) out |= *dst & ~((1u << (8u - (b0 - 8u * y))) - 1u);
            *dst = (unsigned char)out;
        }
    }
}

__global__ void __launch_bounds__(256)
framer_payload_kernel(const unsigned *__restrict__ D, const FramerState *S, const FramerJob *__restrict__ jobs,
                      unsigned char *__restrict__ pool)
{
    framer_payload_body(D, S, jobs, pool);
}

__global__ void __launch_bounds__(256)
framer_payload_batch_kernel(const unsigned *__restrict__ D, long long words_stride, const FramerState *S,
                            const FramerJob *__restrict__ jobs, long long rec_stride, unsigned char *__restrict__ pool,
                            long long pool_stride)
{
    const int s = blockIdx.y;
    framer_payload_body(D + (long long)s * words_stride, S + s, jobs + (long long)s * rec_stride, pool + (long long)s * pool_stride);
}

// after a fetch: drop the delivered messages, move the open packet to the front of the pool
__global__ void __launch_bounds__(256)
framer_compact_kernel(FramerState *S, unsigned char *pool)
{
    __shared__ unsigned char keep[4096];
    const bool open = S->mode == FR_HAVE_HEADER;
    const unsigned off = S->open_off, nb = open ? ((unsigned)S->bits_done + 7u) >> 3 : 0u;
    for (unsigned i = threadIdx.x; i < nb; i += 256) keep[i] = pool[off + i];
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nb; i += 256) pool[i] = keep[i];
    if (threadIdx.x == 0) {
        S->msg_count = 0;
        S->open_off = 0;
        S->pool_used = open ? (unsigned)S->pktlen : 0u;
    }
}

}  // namespace grhip

struct grhip_framer_sink_1 : HandleBase {
    DevBuf d_state, d_F, d_D, d_jobs, d_msgs, d_pool, d_recs, d_segs;
    size_t msg_bound = 0, pool_bound = 4096;     // upper bounds of what un-fetched calls can have produced
    long long seg_items = 0;                     // items per segment of the parallel walk; 0 = chosen per call
    // host copy of the last fetch
    std::vector<FramerMsg> h_msgs;
    std::vector<unsigned char> h_pool;
    size_t next_msg = 0;

    int grow(DevBuf &b, size_t keep_bytes, size_t want, hipStream_t st)
    {
        if (want <= b.cap) return GRHIP_OK;
        DevBuf nb;
        int rc = nb.reserve(want * 2);
        if (rc) return rc;
        if (b.p && keep_bytes) GRHIP_HIP(hipMemcpyAsync(nb.p, b.p, std::min(keep_bytes, b.cap), hipMemcpyDeviceToDevice, st));
        GRHIP_HIP(hipStreamSynchronize(st));
        b.release();
        b = nb;
        return GRHIP_OK;
    }
};

int grhip_framer_sink_1_create(grhip_framer_sink_1 **h, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    auto *b = new (std::nothrow) grhip_framer_sink_1();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    int rc = b->init_device(device);
    if (!rc) rc = b->d_state.reserve(sizeof(FramerState));
    if (!rc && hipMemset(b->d_state.p, 0, sizeof(FramerState)) != hipSuccess) rc = fail(GRHIP_ERUNTIME, "hipMemset failed");   // enter_search(), .cc:84
    if (rc) { grhip_framer_sink_1_destroy(b); return rc; }
    *h = b;
    return GRHIP_OK;
}

void grhip_framer_sink_1_destroy(grhip_framer_sink_1 *h)
{
    if (!h) return;
    (void)h->bind();
    h->d_state.release(); h->d_F.release(); h->d_D.release(); h->d_jobs.release(); h->d_msgs.release(); h->d_pool.release(); h->d_recs.release(); h->d_segs.release();
    h->destroy_base();
    delete h;
}

int grhip_framer_sink_1_set_segment_items(grhip_framer_sink_1 *h, long long items)
{
    if (!h || items < 0) return fail(GRHIP_EINVAL, "bad argument");
    h->seg_items = items;
    return GRHIP_OK;
}

int grhip_framer_sink_1_work_device(grhip_framer_sink_1 *h, int noutput_items, const unsigned char *d_in, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative item count");
    if (noutput_items == 0) return 0;
    if (!d_in) return fail(GRHIP_EINVAL, "null buffer");
    int rc = h->bind();
    if (rc) return rc;
    hipStream_t st = h->pick(stream);
    const long long n = noutput_items, nwords = (n + 31) >> 5, nwp = nwords + 2;
    if ((rc = h->d_F.reserve((size_t)nwp * 4))) return rc;
    if ((rc = h->d_D.reserve((size_t)nwp * 4))) return rc;
    if ((rc = h->d_jobs.reserve((size_t)(n / 32 + 8) * sizeof(FramerJob)))) return rc;
    // every packet costs at least 32 items; payload bytes cost 8 items each; one packet of up to 4095 bytes may be open
    if ((rc = h->grow(h->d_msgs, h->msg_bound * sizeof(FramerMsg), (h->msg_bound + (size_t)n / 32 + 8) * sizeof(FramerMsg), st))) return rc;
    if ((rc = h->grow(h->d_pool, h->pool_bound, h->pool_bound + (size_t)n / 8 + 4096 + 16, st))) return rc;
    h->msg_bound += (size_t)n / 32 + 2;
    h->pool_bound += (size_t)n / 8 + 4096;
    if (h->pool_bound > 0xf0000000ull) return fail(GRHIP_EINVAL, "framer_sink_1: fetch the messages before queueing more than 4 GiB of payload");

    const unsigned pack_blocks = (unsigned)std::min<long long>((nwp + 255) / 256, 8192);
    hipLaunchKernelGGL(framer_pack_kernel, dim3(pack_blocks), dim3(256), 0, st, d_in, n, h->d_F.as<unsigned>(),
                       h->d_D.as<unsigned>(), nwp);
    // items per segment: the fix-up costs about 0.1 us per segment it can take over wholesale, a segment's own
    // walk about 0.63 us per 1000 items, so the two balance near seg = sqrt(160 n) (tools/bench_framer.py)
    long long seg = 4096;
    while (4 * seg * seg <= 160 * n && seg < (1ll << 20)) seg <<= 1;
    if (h->seg_items > 0) seg = std::max<long long>(64, h->seg_items);      // grhip_framer_sink_1_set_segment_items
    const long long nseg = (n + seg - 1) / seg;
    if (nseg >= 4) {
        const size_t reccap = (size_t)seg / 32 + 2;
        if ((rc = h->d_recs.reserve((size_t)nseg * reccap * sizeof(FrRec)))) return rc;
        if ((rc = h->d_segs.reserve((size_t)nseg * sizeof(FrSeg)))) return rc;
        hipLaunchKernelGGL(framer_segwalk_kernel, dim3((unsigned)nseg), dim3(64), 0, st, h->d_F.as<unsigned>(),
                           h->d_D.as<unsigned>(), (unsigned)n, (unsigned)seg, (unsigned)reccap, h->d_recs.as<FrRec>(),
                           h->d_segs.as<FrSeg>());
        hipLaunchKernelGGL(framer_fixup_kernel, dim3(1), dim3(64), 0, st, h->d_F.as<unsigned>(), h->d_D.as<unsigned>(),
                           (unsigned)n, (unsigned)seg, (unsigned)nseg, (unsigned)reccap, h->d_recs.as<FrRec>(),
                           h->d_segs.as<FrSeg>(), h->d_state.as<FramerState>(), h->d_msgs.as<FramerMsg>(),
                           h->d_jobs.as<FramerJob>());
        hipLaunchKernelGGL(framer_emit_kernel, dim3((unsigned)nseg), dim3(256), 0, st, (unsigned)n, (unsigned)reccap,
                           h->d_recs.as<FrRec>(), h->d_segs.as<FrSeg>(), h->d_state.as<FramerState>(),
                           h->d_msgs.as<FramerMsg>(), h->d_jobs.as<FramerJob>());
    } else {
        hipLaunchKernelGGL(framer_walk_kernel, dim3(1), dim3(64), 0, st, h->d_F.as<unsigned>(), h->d_D.as<unsigned>(),
                           (unsigned)n, h->d_state.as<FramerState>(), h->d_msgs.as<FramerMsg>(), h->d_jobs.as<FramerJob>());
    }
    const unsigned pay_blocks = (unsigned)std::min<long long>(n / 256 + 1, 2048);
    hipLaunchKernelGGL(framer_payload_kernel, dim3(pay_blocks), dim3(256), 0, st, h->d_D.as<unsigned>(),
                       h->d_state.as<FramerState>(), h->d_jobs.as<FramerJob>(), h->d_pool.as<unsigned char>());
    GRHIP_HIP(hipGetLastError());
    return noutput_items;                        // a sink: consumes everything (.cc:189)
}

// brings the complete messages of all calls so far to the host and empties the device queue
static int framer_fetch(grhip_framer_sink_1 *h, hipStream_t st)
{
    FramerState s;
    GRHIP_HIP(hipMemcpyAsync(&s, h->d_state.p, sizeof(s), hipMemcpyDeviceToHost, st));
    GRHIP_HIP(hipStreamSynchronize(st));
    // drop what the caller has already popped
    h->h_msgs.erase(h->h_msgs.begin(), h->h_msgs.begin() + h->next_msg);
    h->next_msg = 0;
    if (s.msg_count) {
        std::vector<FramerMsg> m(s.msg_count);
        std::vector<unsigned char> p(s.pool_used ? s.pool_used : 1);
        GRHIP_HIP(hipMemcpyAsync(m.data(), h->d_msgs.p, m.size() * sizeof(FramerMsg), hipMemcpyDeviceToHost, st));
        if (s.pool_used) GRHIP_HIP(hipMemcpyAsync(p.data(), h->d_pool.p, s.pool_used, hipMemcpyDeviceToHost, st));
        GRHIP_HIP(hipStreamSynchronize(st));
        // append: payloads re-based onto the host pool
        if (h->h_msgs.empty()) h->h_pool.clear();
        const unsigned base = (unsigned)h->h_pool.size();
        h->h_msgs.reserve(h->h_msgs.size() + m.size());
        h->h_pool.reserve(h->h_pool.size() + p.size());
        for (auto &r : m) {
            FramerMsg q = r;
            q.off = (unsigned)h->h_pool.size();
            h->h_pool.insert(h->h_pool.end(), p.begin() + r.off, p.begin() + r.off + r.len);
            h->h_msgs.push_back(q);
        }
        (void)base;
    }
    if (h->d_pool.p) {
        hipLaunchKernelGGL(framer_compact_kernel, dim3(1), dim3(256), 0, st, h->d_state.as<FramerState>(),
                           h->d_pool.as<unsigned char>());
        GRHIP_HIP(hipGetLastError());
    }
    h->msg_bound = 0;
    h->pool_bound = 4096;
    return GRHIP_OK;
}

int grhip_framer_sink_1_message_count(grhip_framer_sink_1 *h, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    int rc = h->bind();
    if (rc) return rc;
    if ((rc = framer_fetch(h, h->pick(stream)))) return rc;
    return (int)(h->h_msgs.size() - h->next_msg);
}

int grhip_framer_sink_1_pop(grhip_framer_sink_1 *h, int *whitener_offset, unsigned char *payload, int capacity)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (h->next_msg >= h->h_msgs.size()) return fail(GRHIP_EINVAL, "framer_sink_1: no message (call message_count first)");
    const FramerMsg &m = h->h_msgs[h->next_msg];
    if ((int)m.len > capacity || (m.len && !payload)) return fail(GRHIP_EINVAL, "framer_sink_1: payload buffer too small (4096 always fits)");
    if (whitener_offset) *whitener_offset = (int)m.woff;
    if (m.len) memcpy(payload, h->h_pool.data() + m.off, m.len);
    ++h->next_msg;
    return (int)m.len;
}

// pops up to max_msgs messages at once: whitener offsets, lengths, payloads back to back; stops before a message
// whose payload would not fit.  Returns the number popped.
int grhip_framer_sink_1_drain(grhip_framer_sink_1 *h, int max_msgs, int *whitener_offsets, int *lengths,
                              unsigned char *payload, size_t payload_capacity)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (max_msgs < 0 || (max_msgs && (!whitener_offsets || !lengths))) return fail(GRHIP_EINVAL, "bad arguments");
    int k = 0;
    size_t used = 0;
    while (k < max_msgs && h->next_msg < h->h_msgs.size()) {
        const FramerMsg &m = h->h_msgs[h->next_msg];
        if (used + m.len > payload_capacity || (m.len && !payload)) break;
        whitener_offsets[k] = (int)m.woff;
        lengths[k] = (int)m.len;
        if (m.len) memcpy(payload + used, h->h_pool.data() + m.off, m.len);
        used += m.len;
        ++k;
        ++h->next_msg;
    }
    return k;
}

// ---- multi-capture entry: n_streams captures framed in one go (grid.y / blockIdx.x = stream) --------------
struct grhip_framer_sink_1_batch : HandleBase {
    int S = 0;
    long long max_items = 0, words_stride = 0, rec_stride = 0, pool_stride = 0;
    DevBuf d_state, d_F, d_D, d_jobs, d_msgs, d_pool;
    std::vector<FramerState> h_state;
    std::vector<FramerMsg> h_msgs;
    std::vector<unsigned char> h_pool;
    bool fetched = false;
};

int grhip_framer_sink_1_batch_create(grhip_framer_sink_1_batch **h, int n_streams, size_t max_items_per_stream, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (n_streams < 1 || max_items_per_stream < 1 || max_items_per_stream >= (1ull << 31))
        return fail(GRHIP_EINVAL, "framer_sink_1_batch: bad stream count / capture length");
    auto *b = new (std::nothrow) grhip_framer_sink_1_batch();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    b->S = n_streams;
    b->max_items = (long long)max_items_per_stream;
    b->words_stride = ((b->max_items + 31) >> 5) + 2;            // two zero words behind every stream
    b->rec_stride = b->max_items / 32 + 8;                        // a packet costs at least 32 items
    b->pool_stride = ((b->max_items / 8 + 4096 + 16 + 15) / 16) * 16;
    const size_t S = (size_t)n_streams;
    int rc = b->init_device(device);
    if (!rc) rc = b->d_state.reserve(S * sizeof(FramerState));
    if (!rc) rc = b->d_F.reserve(S * b->words_stride * 4);
    if (!rc) rc = b->d_D.reserve(S * b->words_stride * 4);
    if (!rc) rc = b->d_jobs.reserve(S * b->rec_stride * sizeof(FramerJob));
    if (!rc) rc = b->d_msgs.reserve(S * b->rec_stride * sizeof(FramerMsg));
    if (!rc) rc = b->d_pool.reserve(S * b->pool_stride);
    if (rc) { grhip_framer_sink_1_batch_destroy(b); return rc; }
    *h = b;
    return GRHIP_OK;
}

void grhip_framer_sink_1_batch_destroy(grhip_framer_sink_1_batch *h)
{
    if (!h) return;
    (void)h->bind();
    h->d_state.release(); h->d_F.release(); h->d_D.release(); h->d_jobs.release(); h->d_msgs.release(); h->d_pool.release();
    h->destroy_base();
    delete h;
}

int grhip_framer_sink_1_batch_run_device(grhip_framer_sink_1_batch *h, const unsigned char *d_in, size_t stream_stride_items,
                                         const int *d_nitems, int nitems_stride, size_t n_items_max, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (!d_in) return fail(GRHIP_EINVAL, "null buffer");
    if ((long long)n_items_max > h->max_items) return fail(GRHIP_EINVAL, "n_items_max exceeds max_items_per_stream");
    int rc = h->bind();
    if (rc) return rc;
    hipStream_t st = h->pick(stream);
    h->fetched = false;
    if (n_items_max == 0) { GRHIP_HIP(hipMemsetAsync(h->d_state.p, 0, (size_t)h->S * sizeof(FramerState), st)); return GRHIP_OK; }
    const unsigned pack_blocks = (unsigned)std::min<long long>((h->words_stride + 255) / 256, 1024);
    hipLaunchKernelGGL(framer_pack_batch_kernel, dim3(pack_blocks, (unsigned)h->S), dim3(256), 0, st, d_in, (long long)stream_stride_items,
                       d_nitems, nitems_stride, (long long)n_items_max, h->d_F.as<unsigned>(), h->d_D.as<unsigned>(), h->words_stride);
    hipLaunchKernelGGL(framer_walk_batch_kernel, dim3((unsigned)h->S), dim3(64), 0, st, h->d_F.as<unsigned>(), h->d_D.as<unsigned>(),
                       h->words_stride, d_nitems, nitems_stride, (long long)n_items_max, h->d_state.as<FramerState>(),
                       h->d_msgs.as<FramerMsg>(), h->d_jobs.as<FramerJob>(), h->rec_stride);
    const unsigned pay_blocks = (unsigned)std::min<long long>((long long)n_items_max / 4096 + 1, 64);
    hipLaunchKernelGGL(framer_payload_batch_kernel, dim3(pay_blocks, (unsigned)h->S), dim3(256), 0, st, h->d_D.as<unsigned>(),
                       h->words_stride, h->d_state.as<FramerState>(), h->d_jobs.as<FramerJob>(), h->rec_stride,
                       h->d_pool.as<unsigned char>(), h->pool_stride);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

int grhip_framer_sink_1_batch_fetch(grhip_framer_sink_1_batch *h, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    int rc = h->bind();
    if (rc) return rc;
    hipStream_t st = h->pick(stream);
    const size_t S = (size_t)h->S;
    h->h_state.resize(S);
    GRHIP_HIP(hipMemcpyAsync(h->h_state.data(), h->d_state.p, S * sizeof(FramerState), hipMemcpyDeviceToHost, st));
    GRHIP_HIP(hipStreamSynchronize(st));
    h->h_msgs.assign(S * (size_t)h->rec_stride, FramerMsg{});
    h->h_pool.assign(S * (size_t)h->pool_stride, 0);
    long long total = 0;
    for (size_t s = 0; s < S; ++s) {
        const FramerState &fs = h->h_state[s];
        if (!fs.msg_count) continue;
        total += fs.msg_count;
        GRHIP_HIP(hipMemcpyAsync(h->h_msgs.data() + s * h->rec_stride, h->d_msgs.as<FramerMsg>() + s * h->rec_stride,
                                 (size_t)fs.msg_count * sizeof(FramerMsg), hipMemcpyDeviceToHost, st));
        if (fs.pool_used)
            GRHIP_HIP(hipMemcpyAsync(h->h_pool.data() + s * h->pool_stride, h->d_pool.as<unsigned char>() + s * h->pool_stride,
                                     std::min<size_t>(fs.pool_used, (size_t)h->pool_stride), hipMemcpyDeviceToHost, st));
    }
    GRHIP_HIP(hipStreamSynchronize(st));
    h->fetched = true;
    return (int)std::min<long long>(total, 0x7ffffffe);
}

int grhip_framer_sink_1_batch_count(grhip_framer_sink_1_batch *h, int stream_index)
{
    if (!h || !h->fetched) return fail(GRHIP_EINVAL, "framer_sink_1_batch: call fetch first");
    if (stream_index < 0 || stream_index >= h->S) return fail(GRHIP_EINVAL, "stream index out of range");
    return (int)h->h_state[(size_t)stream_index].msg_count;
}

int grhip_framer_sink_1_batch_get(grhip_framer_sink_1_batch *h, int stream_index, int msg_index, int *whitener_offset,
                                  unsigned char *payload, int capacity)
{
    if (!h || !h->fetched) return fail(GRHIP_EINVAL, "framer_sink_1_batch: call fetch first");
    if (stream_index < 0 || stream_index >= h->S) return fail(GRHIP_EINVAL, "stream index out of range");
    const FramerState &fs = h->h_state[(size_t)stream_index];
    if (msg_index < 0 || (unsigned)msg_index >= fs.msg_count) return fail(GRHIP_EINVAL, "message index out of range");
    const FramerMsg &m = h->h_msgs[(size_t)stream_index * h->rec_stride + (size_t)msg_index];
    if ((int)m.len > capacity || (m.len && !payload)) return fail(GRHIP_EINVAL, "payload buffer too small (4096 always fits)");
    if (whitener_offset) *whitener_offset = (int)m.woff;
    if (m.len) memcpy(payload, h->h_pool.data() + (size_t)stream_index * h->pool_stride + m.off, m.len);
    return (int)m.len;
}

int grhip_framer_sink_1_work(grhip_framer_sink_1 *h, int noutput_items, const unsigned char *in)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative item count");
    if (noutput_items == 0) return 0;
    if (!in) return fail(GRHIP_EINVAL, "null buffer");
    int rc = h->bind();
    if (rc) return rc;
    if ((rc = h->stage_in.reserve((size_t)noutput_items))) return rc;
    hipStream_t st = h->own_stream;
    GRHIP_H2D(h, h->stage_in.p, in, (size_t)noutput_items, st);
    rc = grhip_framer_sink_1_work_device(h, noutput_items, h->stage_in.as<unsigned char>(), st);
    if (rc < 0) return rc;
    GRHIP_HIP(hipStreamSynchronize(st));
    return noutput_items;
}
