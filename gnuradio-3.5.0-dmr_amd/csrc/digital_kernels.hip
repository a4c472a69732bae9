// digital_kernels.hip -- gfx950 kernels for the gr-digital tail of the DMR chain:
//   digital_clock_recovery_mm_ff (+ gri_mmse_fir_interpolator),
//   digital_binary_slicer_fb, digital_correlate_access_code_bb.
#include "digital_kernels.h"
#include <type_traits>

#include "device_math.h"
#include "grhip_internal.h"

namespace grhip {

// ===========================================================================
// Mueller & Mueller clock recovery
//   gr-digital/lib/digital_clock_recovery_mm_ff.cc:104-139
//   filter/gri_mmse_fir_interpolator.cc:61-71 (8-tap gr_fir_fff, 129 phases)
// The loop is serial and data dependent: one wavefront per stream, and a
// lone wavefront issues one instruction every ~5 cycles, so the cost of a symbol is the
// NUMBER of instructions between two symbols.  The kernel keeps that number small:
//  * all 64 lanes stage the next MM_CH input floats into LDS with coalesced loads;
//  * the loop state (mu, omega, last sample) is computed redundantly by every lane, the
//    sample position and output count are wave-uniform scalars: no exec-mask juggling;
//  * lanes 0..7 each form ONE tap product of the 8-tap interpolator (two LDS reads per
//    symbol instead of sixteen) and DPP row shifts add them up in the reference's order,
//    gr_fir_fff_generic::filter with N_UNROLL = 4: acc_j = (0 + p_j) + p_{j+4},
//    out = ((acc_0 + acc_1) + acc_2) + acc_3;
//  * outputs are collected in LDS and written out coalesced once per window.
// Every float operation is a single unfused IEEE op in the reference's order: bit-exact.
// ===========================================================================
constexpr int MM_CH = 2048;        // (8 KB: four such waves fit in the LDS a FIR workgroup leaves on a CU)
constexpr int MM_NTAPS = 8;
constexpr int MM_NSTEPS = 128;

__device__ __forceinline__ float mm_slice_mul(float s, float v) { return s < 0 ? -v : v; }     // slice(s) * v, slice = -1 / +1

// lane i <- lane i+n of its row of 16 (zero beyond the row)
template <int N>
__device__ __forceinline__ float row_shl(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}

// One symbol costs its INSTRUCTION COUNT (a lone wavefront issues one instruction every 4-5 cycles),
// so the loop body is kept to what the recurrence needs (~45 instructions, 75 in round 1):
//  * the interpolator phase of the NEXT symbol is formed at the end of the current one (mu is in
//    [0, 1) from then on: no clamp in the loop; the clamp guards the caller's initial mu only);
//  * slice(a) * b is b with its sign flipped by the sign bit of a (a is never -0: a sum that
//    started from +0), two bit operations instead of compare / select / multiply;
//  * the window and end-of-input tests are ONE unsigned compare, the output count a countdown;
//  * outputs collect in a register (lane c of it = symbol c of a group of 64) and go to HBM
//    with one coalesced store per group: no LDS traffic per symbol.
// resume != 0: counts[2s] / counts[2s+1] hold what stream s has produced / consumed so far; the
// call continues from there (in and out are indexed from the stream's start) and adds to them.
// This is how the chain overlaps the FIR of later parts of a capture with the clock recovery of
// earlier ones; the recurrence is the same whatever the chunking.
__global__ void __launch_bounds__(64)
mm_kernel(MMState *__restrict__ state, int noutput_items, int ninput_items, const float *__restrict__ in,
          long long in_stride, float *__restrict__ out, long long out_stride, int *__restrict__ counts,
          const float *__restrict__ mmse_rev, int resume, int *__restrict__ counts_out)
{
    __shared__ float s_in[MM_CH];
    __shared__ float s_taps[MM_NTAPS * (MM_NSTEPS + 1)];

    const int s = blockIdx.x, lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);      // latency-bound: when it shares a SIMD with a throughput kernel, it goes first
    int oo0 = 0, ii0 = 0;
    if (resume) {
        oo0 = __builtin_amdgcn_readfirstlane(counts[2 * s]);
        ii0 = __builtin_amdgcn_readfirstlane(counts[2 * s + 1]);
    }
    // (sample positions count from the stream's start in every launch: a loop that steps back behind the point a resumed
    // launch started from goes on as it would in one launch, and a stream that has stepped before its first sample -- where
    // the reference would read before its buffer -- stays ended)
    const float *__restrict__ x = in + (long long)s * in_stride;
    float *__restrict__ y = out + (long long)s * out_stride + oo0;
    noutput_items -= oo0;

    for (int i = lane; i < MM_NTAPS * (MM_NSTEPS + 1); i += 64) s_taps[i] = mmse_rev[i];

    const MMState st = state[s];
    float mu = st.mu, omega = st.omega, last = st.last_sample;          // identical in every lane
    const float omega_mid = st.omega_mid, gain_omega = st.gain_omega, gain_mu = st.gain_mu;
    const float rel = st.omega_relative_limit;
    int ii = ii0, oo = 0;                             // wave-uniform (kept in SGPRs)
    const int ni = ninput_items - MM_NTAPS;           // .cc:113
    const int k = lane & 7;                           // this lane's tap of the interpolator
    const float *tapcol = &s_taps[k * (MM_NSTEPS + 1)];
    // interpolate(&in[ii], d_mu): imu = (int) rint(mu * NSTEPS)   (gri_mmse_fir_interpolator.cc:64)
    int imu = (int)__builtin_rintf(mu * (float)MM_NSTEPS);
    imu = imu < 0 ? 0 : (imu > MM_NSTEPS ? MM_NSTEPS : imu);
    bool done = !(oo < noutput_items && ii < ni) || ii < 0;

    while (!done) {
        const int base = ii;
        // sixteen independent loads in flight per lane: one load per loop trip would make the lone wave wait for
        // memory latency 64 times per window
        for (int ib = lane; ib < MM_CH; ib += 64 * 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const long long g = (long long)base + ib + 64 * q;
                v[q] = 0.f;
                if (g >= 0 && g < ninput_items) v[q] = x[g];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) s_in[ib + 64 * q] = v[q];
        }
        __syncthreads();
        // symbols may start at ii in [base, min(ni - 1, base + MM_CH - MM_NTAPS)]: one unsigned compare
        const int lim = base + MM_CH - MM_NTAPS;
        const unsigned span = (unsigned)((ni - 1 < lim ? ni - 1 : lim) - base);
        const float *xk = &s_in[k - base];
        while (oo < noutput_items && (unsigned)(ii - base) <= span) {
            const int room = noutput_items - oo;
            const int grp = room < 64 ? room : 64;
            float obuf = 0.f;
            int c = 0;
            do {
                // lanes 0..7: (0 + tap k * sample k) -- the fma rounds the product once and adds an
                // exact zero, i.e. the reference's `acc = 0; acc += t*x` including the sign of a
                // zero product
                const float p = __builtin_fmaf(tapcol[imu], xk[ii], 0.0f);
                const float acc = p + row_shl<4>(p);                          // lanes 0..3: acc_j
                float o = acc + row_shl<1>(acc);
                o = o + row_shl<2>(acc);
                o = o + row_shl<3>(acc);
                const unsigned ob = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, o), 0);
                // (lane select through M0: a second SGPR operand would break the one-scalar-operand rule of gfx9 VALU)
                asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(obuf) : "s"(ob), "s"(c));
                // .cc:120  mm_val = slice(last) * o - slice(o) * last, slice(x) = x < 0 ? -1 : 1.
                // Neither o nor last is ever -0 (sums that started from +0), so multiplying by
                // slice(a) is flipping the sign by a's sign bit; the two products are exact and the
                // subtraction rounds once, as in the reference.
                const unsigned lb = __builtin_bit_cast(unsigned, last);
                const float t1 = __builtin_bit_cast(float, ob ^ (lb & 0x80000000u));
                const float t2 = __builtin_bit_cast(float, lb ^ (ob & 0x80000000u));
                const float mm_val = t1 - t2;
                last = __builtin_bit_cast(float, ob);
                omega = omega + gain_omega * mm_val;                              // .cc:123
                omega = omega_mid + branchless_clip(omega - omega_mid, rel);      // .cc:124
                mu = mu + omega + gain_mu * mm_val;                               // .cc:125
                const float fl = __builtin_floorf(mu);
                ii = __builtin_amdgcn_readfirstlane(ii + (int)fl);                // .cc:127
                mu = mu - fl;                                                     // .cc:128
                imu = (int)__builtin_rintf(mu * (float)MM_NSTEPS);                // next symbol's phase: mu is in [0, 1)
                ++c;
            } while (c < grp && (unsigned)(ii - base) <= span);
            if (lane < c) y[oo + lane] = obuf;
            oo += c;
        }
        done = !(oo < noutput_items && ii < ni);
        if (ii < 0) done = true;     // the reference would read before its buffer here
        __syncthreads();
    }
    if (lane == 0) {
        MMState so = st;
        so.mu = mu; so.omega = omega; so.last_sample = last;
        state[s] = so;
        int *co = counts_out ? counts_out : counts;       // (the chain keeps the totals of every time slice: a later stage reads
        co[2 * s + 0] = oo0 + oo;                         // slice c's while slice c + 1 runs)
        co[2 * s + 1] = ii;                    // consume_each(ii)
    }
}

// ---------------------------------------------------------------------------
// mm_rows_kernel: EIGHT captures per wavefront, one per group of 8 lanes (the chain's big batches).
// The recurrence of one capture occupies 8 lanes (one tap product each) and a lone wave issues an instruction every
// 4-5 cycles whatever its lanes hold, so the seven other groups of mm_kernel's wave repeated the same work: here each
// group of 8 lanes runs its own capture -- same instructions, same order of the float operations (bit-exact), eight
// symbols per pass.  What was wave-uniform scalar state (sample position, counts) is group-uniform vector state; a
// group that has to wait (window exhausted) or has finished is masked.
//  * samples: a ring of MMR_RING floats per capture in LDS, topped up half a ring at a time; the next half is requested
//    into registers as soon as the previous one is accepted, so a top-up is an LDS write -- and the request is half a ring
//    ahead of its use: 512 samples = 51 symbols = 9 us at 10 samples per symbol with the ring of 1024 (36 KB of LDS per
//    wave: four waves per CU), half that with the ring of 512 (20 KB: more waves per CU).  Beside a kernel that keeps HBM
//    busy a request takes microseconds: 2048 captures beside the FIR run in 72 / 59 / 53 ms with rings of 256 / 512 /
//    1024.  A group accepts a chunk once it has left the older half of its ring, and every group that can accepts when
//    any group must (the wave leaves the symbol loop at most once per half ring); a position outside the ring (a jump
//    of the loop) re-seeds the ring there;
//  * the 8-tap sum is formed with the same DPP row shifts (the two groups of a 16-lane row do not meet in the lanes that
//    count), the result goes to the group's lanes with a quad broadcast and a masked row shift;
//  * outputs collect in one register per lane, a group stores 8 symbols (32 bytes) at a time.
// in / out are indexed from the stream's start (resume != 0: counts[] hold what has been produced / consumed so far);
// the caller guarantees 16-byte aligned rows with at least 3 floats of slack behind ninput_items (the chain's rows).
// ---------------------------------------------------------------------------
constexpr int MMR_RL = 8, MMR_ROWS = 64 / MMR_RL;

template <int MMR_RING>
__global__ void __launch_bounds__(64)
mm_rows_kernel(MMState *__restrict__ state, int n_streams, int noutput_items, int ninput_items,
               const float *__restrict__ in, long long in_stride, float *__restrict__ out, long long out_stride,
               int *__restrict__ counts, const float *__restrict__ mmse_rev, int resume, int *__restrict__ counts_out)
{
    constexpr int MMR_CHUNK = MMR_RING / 2;
    __shared__ __attribute__((aligned(16))) float s_ring[MMR_ROWS * MMR_RING];
    __shared__ float s_taps[MM_NTAPS * (MM_NSTEPS + 1)];
    typedef float f4 __attribute__((ext_vector_type(4)));

    const int lane = threadIdx.x, row = lane / MMR_RL, l = lane % MMR_RL;
    const int s = blockIdx.x * MMR_ROWS + row;
    const bool valid = s < n_streams;
    const int sc = valid ? s : n_streams - 1;
    __builtin_amdgcn_s_setprio(3);
    for (int i = lane; i < MM_NTAPS * (MM_NSTEPS + 1); i += 64) s_taps[i] = mmse_rev[i];

    int oo = 0, ii = 0;                               // produced (stored) / consumed so far, from the stream's start
    if (resume) { oo = counts[2 * sc]; ii = counts[2 * sc + 1]; }
    const float *__restrict__ x = in + (long long)sc * in_stride;
    float *__restrict__ y = out + (long long)sc * out_stride;
    const MMState st = state[sc];
    float mu = st.mu, omega = st.omega, last = st.last_sample;          // identical in the lanes of a group
    const float omega_mid = st.omega_mid, gain_omega = st.gain_omega, gain_mu = st.gain_mu;
    const float rel = st.omega_relative_limit;
    const int ni = ninput_items - MM_NTAPS;           // .cc:113
    const float *tapcol = &s_taps[l * (MM_NSTEPS + 1)];
    float *ring = &s_ring[row * MMR_RING];
    int imu = (int)__builtin_rintf(mu * (float)MM_NSTEPS);
    imu = imu < 0 ? 0 : (imu > MM_NSTEPS ? MM_NSTEPS : imu);
    bool fin = !valid || !(oo < noutput_items && ii < ni) || ii < 0;
    int lo = 0, hi = 0;                               // the ring holds samples [lo, hi)
    constexpr int NQ = MMR_CHUNK / (4 * MMR_RL);      // 16-byte loads per lane and chunk
    f4 pf[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) pf[q] = f4{0.f, 0.f, 0.f, 0.f};

    auto load_chunk = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int g = h + 32 * q + 4 * l;
            pf[q] = f4{0.f, 0.f, 0.f, 0.f};
            if (g < ninput_items) pf[q] = *reinterpret_cast<const f4 *>(x + g);
        }
    };
    if (!fin) {
        lo = hi = ii & ~(MMR_CHUNK - 1);
        load_chunk(hi);
    }
    __syncthreads();                                  // tap table

    for (;;) {
        // ---- top-up: every group that has left the older half of its ring takes the chunk it has requested
        if (!fin) {
            if (ii < lo || ii >= hi + MMR_CHUNK) {    // outside the ring and the requested chunk: start again there
                lo = hi = ii & ~(MMR_CHUNK - 1);
                load_chunk(hi);
            }
            if (ii >= hi - MMR_CHUNK) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    *reinterpret_cast<f4 *>(&ring[(hi + 32 * q + 4 * l) & (MMR_RING - 1)]) = pf[q];
                hi += MMR_CHUNK;
                lo = lo > hi - MMR_RING ? lo : hi - MMR_RING;
                load_chunk(hi);
            }
        }
        __syncthreads();                              // (one wave: orders the ring writes before the reads below)
        // a symbol may start at ii in [lo, min(hi - 8, ni - 1)]: ONE unsigned compare per symbol
        const int lim = hi - MM_NTAPS < ni - 1 ? hi - MM_NTAPS : ni - 1;
        const unsigned span = (unsigned)(lim - lo);
        const bool act = !fin && lim >= lo && (unsigned)(ii - lo) <= span;
        if (!__any(act)) {
            if (!__any(!fin)) break;
            continue;
        }
        // every active group takes a symbol per pass, so the count of symbols waiting in obuf is wave-uniform; groups of
        // 8 (one store per group), single symbols while some capture is within 8 outputs of its limit
        const int K = __any(act && noutput_items - oo < MMR_RL) ? 1 : MMR_RL;
        if (act) {
            float obuf = 0.f;
            // one symbol of every active group; returns whether all of them may go on (wave-uniform)
            auto step = [&](int c) __attribute__((always_inline)) -> bool {
                // lanes 0..7 of the group: (0 + tap k * sample k), see mm_kernel
                const float p = __builtin_fmaf(tapcol[imu], ring[(ii + l) & (MMR_RING - 1)], 0.0f);
                const float acc = p + row_shl<4>(p);                          // lanes 0..3 of the group: acc_j
                float o = acc + row_shl<1>(acc);
                o = o + row_shl<2>(acc);
                o = o + row_shl<3>(acc);                                      // lane 0 of the group
                // to the 8 lanes of the group: lane 0 of every quad to its quad, then quads 0 / 2 to quads 1 / 3
                int ob = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o), 0x00, 0xf, 0xf, true);
                ob = __builtin_amdgcn_update_dpp(ob, ob, 0x114, 0xf, 0xa, false);
                obuf = l == c ? __builtin_bit_cast(float, ob) : obuf;
                const unsigned ou = (unsigned)ob, lb = __builtin_bit_cast(unsigned, last);
                const float t1 = __builtin_bit_cast(float, ou ^ (lb & 0x80000000u));     // slice(last) * o
                const float t2 = __builtin_bit_cast(float, lb ^ (ou & 0x80000000u));     // slice(o) * last
                const float mm_val = t1 - t2;                                     // .cc:120
                last = __builtin_bit_cast(float, ob);
                omega = omega + gain_omega * mm_val;                              // .cc:123
                omega = omega_mid + branchless_clip(omega - omega_mid, rel);      // .cc:124
                mu = mu + omega + gain_mu * mm_val;                               // .cc:125
                const float fl = __builtin_floorf(mu);
                ii += (int)fl;                                                    // .cc:127
                mu = mu - fl;                                                     // .cc:128
                imu = (int)__builtin_rintf(mu * (float)MM_NSTEPS);
                return __all((unsigned)(ii - lo) <= span);
            };
            bool inw = true;
            do {
                int c = 1;                            // symbols in obuf after the passes below (wave-uniform)
                if (K == 1) {
                    inw = step(0);
                } else {
                    // (unrolled: the lane that keeps symbol c is a constant mask, the pass count no loop variable)
                    do {
                        inw = step(0); if (!inw) break; ++c;
                        inw = step(1); if (!inw) break; ++c;
                        inw = step(2); if (!inw) break; ++c;
                        inw = step(3); if (!inw) break; ++c;
                        inw = step(4); if (!inw) break; ++c;
                        inw = step(5); if (!inw) break; ++c;
                        inw = step(6); if (!inw) break; ++c;
                        inw = step(7);
                    } while (false);
                }
                if (l < c) y[oo + l] = obuf;
                oo += c;
                if (!(oo < noutput_items)) { fin = true; inw = false; }
                // K was chosen before this loop: once a capture has fewer than 8 outputs of room left, go back for
                // single symbols (eight more steps would store past y[noutput_items - 1] and leave oo beyond the limit
                // the reference stops at, .cc:113)
                if (K != 1 && noutput_items - oo < MMR_RL) inw = false;
                inw = __all(inw);
            } while (inw);
            // left the ring, or the stream: .cc:113 (ii < ni), and the reference would read before its buffer at ii < 0
            if (!(ii < ni) || ii < 0) fin = true;
        }
    }
    if (valid && l == 0) {
        MMState so = st;
        so.mu = mu; so.omega = omega; so.last_sample = last;
        state[s] = so;
        int *co = counts_out ? counts_out : counts;
        co[2 * s + 0] = oo;
        co[2 * s + 1] = ii;                           // consume_each(ii)
    }
}

// ---------------------------------------------------------------------------
// mm_pairs_kernel: THIRTY-TWO captures per wavefront, two lanes each (the chain's biggest batches).
// The loop is a latency chain -- a symbol costs what its ~50 dependent instructions and one LDS round trip cost,
// whatever the lanes hold -- so the batch's clock recovery takes (symbols per capture) x (time per pass) however many
// captures a wave carries, and what a design can choose is how many SIMDs and how much LDS that time occupies beside
// the FIR.  mm_rows_kernel (8 lanes per capture: one tap product per lane, DPP sum) needs a wave per 8 captures and
// 4 KB of ring per capture to ride out the memory latency beside a kernel that keeps HBM's queues full (several us):
// 2048 captures = 256 waves and 9 MB of LDS = 64 CUs the FIR does not get.  Here:
//  * lane h of a pair forms taps 4h .. 4h+3 of the interpolator from ONE 16-byte tap read and four sample reads, the
//    pair's halves meet in one DPP quad swap per accumulator: acc_j = p_j + p_{j+4}, o = ((acc_0 + acc_1) + acc_2) +
//    acc_3 -- the reference's order (gr_fir_fff_generic::filter, N_UNROLL = 4), every operation unfused: bit-exact;
//    lane 0 of the pair hands o to both, which then run the recurrence redundantly;
//  * samples: a ring of MMP_R per capture in LDS, laid out [position][capture] (a pass reads 64 different banks' worth:
//    conflict-free whatever the captures' positions), eight mirrored rows behind the end so a symbol's reads never wrap;
//  * the latency is ridden out in REGISTERS: a FIFO of MMP_NQ chunks of MMP_C samples per capture, requested
//    MMP_NQ chunks ahead (448 samples = 45 symbols) with hand-issued loads and hand-counted waits (vmcnt counts in
//    order; the compiler's bookkeeping cannot follow registers that are loaded in one trip of a loop and used many
//    trips later).  52 KB of the wave's 128 KB of registers hold what mm_rows_kernel holds in LDS;
//  * every capture of the wave accepts its next chunk in the same pass (the FIFO's slots are register NAMES, the same
//    for all lanes); positions are per capture.  Captures drift apart by their symbol rates: a capture that has no room
//    when another must accept (184 samples of slack) makes the wave start its FIFO again at every capture's own position;
//  * outputs collect in LDS and leave as one 16-byte store per lane every eight passes (a store per pass would fill the
//    memory counter the FIFO is counted with).
// 2048 captures = 64 waves = one per SIMD of 16 CUs.  in / out as for mm_rows_kernel (rows 16-byte aligned, the rows of a
// wave within 2 GB: checked by launch_mm).
// ---------------------------------------------------------------------------
constexpr int MMP_CAPS = 32, MMP_R = 256, MMP_C = 64, MMP_NQ = 7;
constexpr int MMP_NL = MMP_C / 8;             // 16-byte loads per lane and chunk

__global__ void __launch_bounds__(64)
mm_pairs_kernel(MMState *__restrict__ state, int n_streams, int noutput_items, int ninput_items,
                const float *__restrict__ in, long long in_stride, float *__restrict__ out, long long out_stride,
                int *__restrict__ counts, const float *__restrict__ mmse_rev, int resume, int *__restrict__ counts_out)
{
    __shared__ __attribute__((aligned(16))) float s_ring[(MMP_R + MM_NTAPS) * MMP_CAPS];
    __shared__ __attribute__((aligned(16))) float s_taps[(MM_NSTEPS + 1) * MM_NTAPS];      // [phase][tap]
    __shared__ __attribute__((aligned(16))) float s_ob[MMP_CAPS * 8];
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int lane = threadIdx.x, c = lane >> 1, h = lane & 1;
    const int s0 = blockIdx.x * MMP_CAPS, s = s0 + c;
    const bool valid = s < n_streams;
    const int sc = valid ? s : n_streams - 1;
    __builtin_amdgcn_s_setprio(3);
    for (int i = lane; i < MM_NTAPS * (MM_NSTEPS + 1); i += 64) {
        const int k = i / (MM_NSTEPS + 1), m = i - k * (MM_NSTEPS + 1);
        s_taps[m * MM_NTAPS + k] = mmse_rev[i];
    }
    int oo = 0, ii = 0;
    if (resume) { oo = counts[2 * sc]; ii = counts[2 * sc + 1]; }
    const MMState st = state[sc];
    float mu = st.mu, omega = st.omega, last = st.last_sample;          // identical in the lanes of a pair
    const float omega_mid = st.omega_mid, gain_omega = st.gain_omega, gain_mu = st.gain_mu;
    const float rel = st.omega_relative_limit;
    const int ni = ninput_items - MM_NTAPS;           // .cc:113
    int imu = (int)__builtin_rintf(mu * (float)MM_NSTEPS);
    imu = imu < 0 ? 0 : (imu > MM_NSTEPS ? MM_NSTEPS : imu);
    bool live = valid && oo < noutput_items && ii < ni && ii >= 0;

    // the wave's rows through two buffer descriptors (32-bit offsets; reads past the last row's data return zeros)
    const int nrow = n_streams - s0 < MMP_CAPS ? n_streams - s0 : MMP_CAPS;
    const unsigned long long xb = (unsigned long long)(in + (long long)s0 * in_stride);
    const unsigned long long yb = (unsigned long long)(out + (long long)s0 * out_stride);
    u32x4 rs_in, rs_out;
    rs_in[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xb);
    rs_in[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffffu));
    rs_in[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)((((long long)(nrow - 1) * in_stride + ninput_items + 3) & ~3ll) * 4));
    rs_in[3] = 0x00020000u;
    rs_out[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)yb);
    rs_out[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(yb >> 32) & 0xffffu));
    rs_out[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)(((long long)(nrow - 1) * out_stride + noutput_items) * 4));
    rs_out[3] = 0x00020000u;
    const int rin = (int)((long long)(sc - s0) * in_stride * 4);        // this capture's row, bytes from the wave's first
    const int rout = (int)((long long)(sc - s0) * out_stride * 4);

    int lo = 0, hi = 0;                               // the ring holds samples [lo, hi) of the capture
    int wlo = (int)0x80000000, wspan = 0;             // a symbol may start at ii with (unsigned)(ii - wlo) <= wspan
    int q = 0;                                        // FIFO slot of the next chunk (the same for every capture)
    // The FIFO lives in a[0 .. 32 MMP_NQ), named by hand: slot k is a[32k .. 32k+31], register 4i + t of it sample
    // 4i + t of the lane's half chunk.  (As a C++ array the allocator gave every load the same four registers and copied
    // them to the array's home right behind the load -- before the data had landed.  The compiler is told these registers
    // are clobbered by every statement that touches them and has no use for accumulation registers of its own here.)
#define MMP_CLOB \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
    "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
    "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
    "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
    "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223"
    // lane h requests samples [pos + 32 h, pos + 32 h + 32) of its capture's chunk at pos
    auto request = [&](auto slot, int pos) __attribute__((always_inline)) {
        constexpr int B = 32 * decltype(slot)::value;
        int vo = live ? rin + (pos + (MMP_C / 2) * h) * 4 : 0x7ffff000;      // (ended captures: no traffic)
#if defined(GRHIP_DIAG) && defined(GRHIP_MMP_ABL)          // timing-only ablations (results are wrong): 1 no requests, 2 no ring writes,
        if (GRHIP_MMP_ABL & 1) return;                          // 4 every capture reads the wave's first row (two lines per request)
        if (GRHIP_MMP_ABL & 4) vo = live ? (pos + (MMP_C / 2) * h) * 4 : 0x7ffff000;
#endif
#define MMP_LD(i) asm volatile("buffer_load_dwordx4 a[%2:%3], %0, %1, 0 offen offset:%4" :: "v"(vo), "s"(rs_in), "n"(B + 4 * (i)), "n"(B + 4 * (i) + 3), "n"(16 * (i)) : MMP_CLOB)
        MMP_LD(0); MMP_LD(1); MMP_LD(2); MMP_LD(3); MMP_LD(4); MMP_LD(5); MMP_LD(6); MMP_LD(7);
#undef MMP_LD
        static_assert(MMP_NL == 8, "eight loads per lane and chunk");
    };
    // the oldest chunk into the ring, its slot requested again MMP_NQ chunks further on
    auto take = [&](auto slot) __attribute__((always_inline)) {
        constexpr int B = 32 * decltype(slot)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MMP_NL * (MMP_NQ - 1)) : "memory");    // all but the younger slots have landed
#if defined(GRHIP_DIAG) && defined(GRHIP_MMP_ABL)
        if ((GRHIP_MMP_ABL & 2) && live) { hi += MMP_C; lo = lo > hi - MMP_R ? lo : hi - MMP_R; } else
#endif
        if (live) {
            const int pb = (hi + (MMP_C / 2) * h) & (MMP_R - 1);
            // rows pb + j, j = 0 .. 31 (128 bytes apart): two rows per instruction, 256 bytes = one unit of st64 apart
            const unsigned w0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)&s_ring[pb * MMP_CAPS + c];
            const unsigned w1 = w0 + 4 * MMP_CAPS;
#define MMP_ST(j) asm volatile("ds_write2st64_b32 %0, a[%1], a[%2] offset0:%3 offset1:%4" :: "v"(((j) & 1) ? w1 : w0), "n"(B + (j)), "n"(B + (j) + 2), "n"((j) / 2), "n"((j) / 2 + 1) : "memory", MMP_CLOB)
            MMP_ST(0); MMP_ST(1); MMP_ST(4); MMP_ST(5); MMP_ST(8); MMP_ST(9); MMP_ST(12); MMP_ST(13);
            MMP_ST(16); MMP_ST(17); MMP_ST(20); MMP_ST(21); MMP_ST(24); MMP_ST(25); MMP_ST(28); MMP_ST(29);
            if (pb == 0) {                                                     // rows 0..7 again behind the ring's end
                const unsigned m0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)&s_ring[MMP_R * MMP_CAPS + c];
                const unsigned m1 = m0 + 4 * MMP_CAPS;
#define MMP_MR(j) asm volatile("ds_write2st64_b32 %0, a[%1], a[%2] offset0:%3 offset1:%4" :: "v"(((j) & 1) ? m1 : m0), "n"(B + (j)), "n"(B + (j) + 2), "n"((j) / 2), "n"((j) / 2 + 1) : "memory", MMP_CLOB)
                MMP_MR(0); MMP_MR(1); MMP_MR(4); MMP_MR(5);
#undef MMP_MR
            }
#undef MMP_ST
            hi += MMP_C;
            lo = lo > hi - MMP_R ? lo : hi - MMP_R;
        }
        request(slot, hi + (MMP_NQ - 1) * MMP_C);
    };
    auto accept = [&]() __attribute__((always_inline)) {
        switch (q) {
        case 0: take(std::integral_constant<int, 0>()); break;
        case 1: take(std::integral_constant<int, 1>()); break;
        case 2: take(std::integral_constant<int, 2>()); break;
        case 3: take(std::integral_constant<int, 3>()); break;
        case 4: take(std::integral_constant<int, 4>()); break;
        case 5: take(std::integral_constant<int, 5>()); break;
        default: take(std::integral_constant<int, 6>()); break;
        }
        static_assert(MMP_NQ == 7, "one case per FIFO slot");
        q = q + 1 == MMP_NQ ? 0 : q + 1;
    };
    auto seed = [&]() __attribute__((always_inline)) {
        if (live) { hi = ii & ~(MMP_C - 1); lo = hi; }
        request(std::integral_constant<int, 0>(), hi);
        request(std::integral_constant<int, 1>(), hi + MMP_C);
        request(std::integral_constant<int, 2>(), hi + 2 * MMP_C);
        request(std::integral_constant<int, 3>(), hi + 3 * MMP_C);
        request(std::integral_constant<int, 4>(), hi + 4 * MMP_C);
        request(std::integral_constant<int, 5>(), hi + 5 * MMP_C);
        request(std::integral_constant<int, 6>(), hi + 6 * MMP_C);
        q = 0;
    };
    // pending outputs: s_ob[capture][0..7], lane h stores symbols 4h .. 4h+3 of its capture
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float *const ob0 = (lds_float *)&s_ob[c * 8];
    lds_float *obp = ob0;                             // behind the capture's last pending symbol
    auto flush = [&]() __attribute__((always_inline)) {
        const int obn = (int)(obp - ob0);
        if (obn > 0) {
            const f4 v = *reinterpret_cast<const __attribute__((address_space(3))) f4 *>(ob0 + 4 * h);
            const int vo = rout + (oo + 4 * h) * 4;
            if (obn == 8) {
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(v), "v"(vo), "s"(rs_out) : "memory");
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (4 * h + t < obn)
                        asm volatile("buffer_store_dword %0, %1, %2, 0 offen offset:%3" :: "v"(v[t]), "v"(vo), "s"(rs_out), "n"(4 * t) : "memory");
            }
            oo += obn;
            obp = ob0;
            if (!(oo < noutput_items)) live = false;
        }
    };

#if defined(GRHIP_DIAG) && defined(GRHIP_MMP_ABL)
    for (int i = lane; i < (MMP_R + MM_NTAPS) * MMP_CAPS; i += 64) s_ring[i] = 0.f;
#endif
    if (__any(live)) seed();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // tap table written (one wave: LDS operations execute in order)
    int group = 0;                                    // passes since the last flush (wave-uniform)
    int K = __any(live && noutput_items - oo < 8) ? 1 : 8;
    for (;;) {
        // ---- between runs of passes: captures that have ended, the next chunk for all, the window of each
        if (live && (!(ii < ni) || ii < 0)) live = false;      // .cc:113; the reference would read before its buffer at ii < 0
        if (!__any(live)) break;
        if (__any(live && !((unsigned)(ii - wlo) <= (unsigned)wspan))) {
            const bool need = live && ii + MM_NTAPS > hi;
            const bool room = !live || ii >= hi + MMP_C - MMP_R;
            const bool lost = live && (ii < lo || ii - hi >= 2 * MMP_R);
            if (__any(lost) || (__any(need) && !__all(room))) seed();
            else if (__any(need)) accept();
            const int lim = hi - MM_NTAPS < ni - 1 ? hi - MM_NTAPS : ni - 1;
            const bool okw = lim >= lo;
            wlo = okw ? lo : (int)0x80000000;
            wspan = okw ? lim - lo : 0;
            continue;
        }
        // ---- passes: one symbol of every running capture each, until one of them leaves its window or the group is full
        const int budget = K - group;
        const int lane_live = (int)__builtin_ctzll(__ballot(live));        // a capture that runs (there is one)
        int n = 0;
        if (live) {
            do {
                const int p = (ii + 4 * h) & (MMP_R - 1);
                const float *rp = &s_ring[p * MMP_CAPS + c];
                const f4 t = *reinterpret_cast<const f4 *>(&s_taps[imu * MM_NTAPS + 4 * h]);
                const float p0 = __builtin_fmaf(t[0], rp[0], 0.0f);                  // (0 + tap * sample), see mm_kernel
                const float p1 = __builtin_fmaf(t[1], rp[MMP_CAPS], 0.0f);
                const float p2 = __builtin_fmaf(t[2], rp[2 * MMP_CAPS], 0.0f);
                const float p3 = __builtin_fmaf(t[3], rp[3 * MMP_CAPS], 0.0f);
                // lane 0 of the pair: acc_j = p_j + p_{j+4} (the other lane's sums are not used); written out, or hipcc pairs the
                // sums into packed adds behind eight moves.  (s_nop: a DPP operand written by the instruction before)
                float a0, a1, a2, a3;
                asm("s_nop 1\n\tv_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                    : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
                float o = a0 + a1;
                o = o + a2;
                o = o + a3;
                float obf;                            // lane 0's o to both lanes of the pair
                asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "=v"(obf) : "v"(o));
                *obp++ = obf;                         // (both lanes: same word, same value)
                const unsigned ou = __builtin_bit_cast(unsigned, obf), lb = __builtin_bit_cast(unsigned, last);
                const float t1 = __builtin_bit_cast(float, ou ^ (lb & 0x80000000u));     // slice(last) * o
                const float t2 = __builtin_bit_cast(float, lb ^ (ou & 0x80000000u));     // slice(o) * last
                const float mm_val = t1 - t2;                                     // .cc:120
                last = obf;
                omega = omega + gain_omega * mm_val;                              // .cc:123
                omega = omega_mid + branchless_clip(omega - omega_mid, rel);      // .cc:124
                mu = mu + omega + gain_mu * mm_val;                               // .cc:125
                const float fl = __builtin_floorf(mu);
                ii += (int)fl;                                                    // .cc:127
                mu = mu - fl;                                                     // .cc:128
                imu = (int)__builtin_rintf(mu * (float)MM_NSTEPS);
                ++n;
            } while (n < budget && __all((unsigned)(ii - wlo) <= (unsigned)wspan));
        }
        group += __builtin_amdgcn_readlane(n, lane_live);                 // (the passes of the run: n is 0 in ended captures' lanes)
        if (group >= K) {
            flush();
            group = 0;
            K = __any(live && noutput_items - oo < 8) ? 1 : 8;
        }
    }
    flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the FIFO's last requests)
    if (valid && h == 0) {
        MMState so = st;
        so.mu = mu; so.omega = omega; so.last_sample = last;
        state[s] = so;
        int *co = counts_out ? counts_out : counts;
        co[2 * s + 0] = oo;
        co[2 * s + 1] = ii;                           // consume_each(ii)
    }
}
#undef MMP_CLOB

int launch_mm(MMState *state, int n_streams, int noutput_items, int ninput_items, const float *in,
              long long in_stride, float *out, long long out_stride, int *counts, const float *mmse_rev,
              hipStream_t st, int resume, int rows, int *counts_out)
{
    if (n_streams <= 0) return GRHIP_OK;
    if (rows == MMP_CAPS) {
        if ((((uintptr_t)in) & 15) || (in_stride & 3) || (((uintptr_t)out) & 3))
            return fail(GRHIP_EINVAL, "clock recovery, thirty-two captures per wave: rows must be 16-byte aligned");
        if (in_stride * 4 * MMP_CAPS >= (1ll << 31) || out_stride * 4 * MMP_CAPS >= (1ll << 31))
            return fail(GRHIP_EINVAL, "clock recovery, thirty-two captures per wave: the rows of a wave must lie within 2 GB");
        const dim3 grid((n_streams + MMP_CAPS - 1) / MMP_CAPS);
        hipLaunchKernelGGL(mm_pairs_kernel, grid, dim3(64), 0, st, state, n_streams, noutput_items, ninput_items, in, in_stride,
                           out, out_stride, counts, mmse_rev, resume, counts_out);
        GRHIP_HIP(hipGetLastError());
        return GRHIP_OK;
    }
    if (rows) {
        if ((((uintptr_t)in) & 15) || (in_stride & 3))
            return fail(GRHIP_EINVAL, "clock recovery, eight captures per wave: rows must be 16-byte aligned");
        const dim3 grid((n_streams + MMR_ROWS - 1) / MMR_ROWS);
        if (rows >= 1024)
            hipLaunchKernelGGL(mm_rows_kernel<1024>, grid, dim3(64), 0, st, state, n_streams, noutput_items, ninput_items, in,
                               in_stride, out, out_stride, counts, mmse_rev, resume, counts_out);
        else
            hipLaunchKernelGGL(mm_rows_kernel<512>, grid, dim3(64), 0, st, state, n_streams, noutput_items, ninput_items, in,
                               in_stride, out, out_stride, counts, mmse_rev, resume, counts_out);
        GRHIP_HIP(hipGetLastError());
        return GRHIP_OK;
    }
    hipLaunchKernelGGL(mm_kernel, dim3(n_streams), dim3(64), 0, st, state, noutput_items, ninput_items, in,
                       in_stride, out, out_stride, counts, mmse_rev, resume, counts_out);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// pager_slicer_fb (gr-pager/lib/pager_slicer_fb.cc:47-84).  The DC tracker
// d_avg = d_avg*beta + x*alpha is a serial float recurrence (bit-exact: two products and
// one sum, unfused).  One wavefront per stream, windows of 1024 samples: all lanes load
// the window and form x*alpha in parallel, every lane then walks the recurrence out of
// LDS redundantly (two dependent operations per sample), and all lanes take the
// decisions in parallel.
// ===========================================================================
constexpr int PS_CH = 1024;

__global__ void __launch_bounds__(64)
pager_slicer_kernel(float *__restrict__ d_avg, float alpha, float beta, const float *__restrict__ in,
                    long long in_stride, unsigned char *__restrict__ out, long long out_stride, long long n,
                    const int *__restrict__ n_ptr, int n_ptr_stride, int *__restrict__ pos)
{
    __shared__ __attribute__((aligned(16))) float s_x[PS_CH], s_t[PS_CH];
    const int s = blockIdx.x, lane = threadIdx.x;
    if (n_ptr) {
        const long long m = n_ptr[(long long)s * n_ptr_stride];
        n = m < n ? (m < 0 ? 0 : m) : n;
    }
    const float *__restrict__ x = in + (long long)s * in_stride;
    unsigned char *__restrict__ y = out + (long long)s * out_stride;
    float avg = d_avg[s];
    // pos: the items of stream s sliced by earlier calls (the chain's time slices); this call goes on from there
    const long long start = pos ? pos[s] : 0;
    for (long long base = start; base < n; base += PS_CH) {
        const int m = (int)(n - base < PS_CH ? n - base : PS_CH);
        {   // the whole window with all of a lane's loads in flight at once (PS_CH / 64 of them)
            float v[PS_CH / 64];
#pragma unroll
            for (int q = 0; q < PS_CH / 64; ++q) {
                const int i = lane + 64 * q;
                v[q] = 0.f;
                if (i < m) v[q] = x[base + i];
            }
#pragma unroll
            for (int q = 0; q < PS_CH / 64; ++q) {
                const int i = lane + 64 * q;
                s_x[i] = v[q];
                s_t[i] = v[q] * alpha;
            }
        }
        __syncthreads();
        // whole groups of 64 samples: sixteen of the alpha x at a time into registers (one LDS wait per sixteen), the recurrence
        // in registers -- the same two unfused operations per sample, in every lane -- and lane l keeps the average of the
        // group's sample l for its decision (round 2 until then: an LDS read, the two operations and an LDS store per sample
        // in one dependent chain: 16.6 ms for 2048 captures of 250 k symbols; this form 12.4 ms; storing the averages to LDS
        // from one lane, sixteen at a time, measured no better than the original)
        typedef float ps_f4 __attribute__((ext_vector_type(4)));
        const int mg = m & ~63;
        for (int g0 = 0; g0 < mg; g0 += 64) {
            float kp = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const ps_f4 *tp = reinterpret_cast<const ps_f4 *>(&s_t[g0 + 16 * c]);
                const ps_f4 ta = tp[0], tb = tp[1], tc = tp[2], td = tp[3];
                const float tv[16] = {ta[0], ta[1], ta[2], ta[3], tb[0], tb[1], tb[2], tb[3], tc[0], tc[1], tc[2], tc[3], td[0], td[1], td[2], td[3]};
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    avg = avg * beta + tv[j];       // .cc:52
                    kp = lane == 16 * c + j ? avg : kp;
                }
            }
            const int i = g0 + lane;
            const float sample = s_x[i] - kp;       // .cc:53
            unsigned char d;
            if (sample > 0) d = (sample > 2.0f) ? 3 : 2;
            else d = (sample < -2.0f) ? 0 : 1;
            y[base + i] = d;
        }
        for (int i = mg; i < m; ++i) {              // the stream's last, partial group
            avg = avg * beta + s_t[i];
            s_t[i] = avg;
        }
        __syncthreads();
        for (int i = mg + lane; i < m; i += 64) {
            const float sample = s_x[i] - s_t[i];   // .cc:53
            unsigned char d;
            if (sample > 0) d = (sample > 2.0f) ? 3 : 2;
            else d = (sample < -2.0f) ? 0 : 1;
            y[base + i] = d;
        }
        __syncthreads();
    }
    if (lane == 0) {
        d_avg[s] = avg;
        if (pos) pos[s] = (int)(n > start ? n : start);
    }
}

int launch_pager_slicer(float *d_avg, int n_streams, float alpha, float beta, const float *in, long long in_stride,
                        unsigned char *out, long long out_stride, long long n, hipStream_t st, const int *n_ptr,
                        int n_ptr_stride, int *pos)
{
    if (n <= 0 || n_streams <= 0) return GRHIP_OK;
    hipLaunchKernelGGL(pager_slicer_kernel, dim3(n_streams), dim3(64), 0, st, d_avg, alpha, beta, in, in_stride, out,
                       out_stride, n, n_ptr, n_ptr_stride, pos);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// gr_unpack_k_bits_bb (general/gr_unpack_k_bits_bb.cc:64-69): output n is bit
// k-1-(n mod k) of input n/k
// ===========================================================================
__global__ void __launch_bounds__(256)
unpack_k_bits_kernel(unsigned k, const unsigned char *__restrict__ in, unsigned char *__restrict__ out, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long long q = i / k;
        const unsigned j = k - 1 - (unsigned)(i - q * k);
        out[i] = (unsigned char)(((unsigned)in[q] >> j) & 1u);
    }
}

// multi-capture form: blockIdx.y = stream; stream s has n_ptr[s * n_ptr_stride] input items (at most n_in_max) and
// gets k times as many output items, whose number goes to n_out[s * n_out_stride]
__global__ void __launch_bounds__(256)
unpack_k_bits_streams_kernel(unsigned k, const unsigned char *__restrict__ in, long long in_stride, unsigned char *__restrict__ out,
                             long long out_stride, long long n_in_max, const int *__restrict__ n_ptr, int n_ptr_stride,
                             int *__restrict__ n_out, int n_out_stride)
{
    const int s = blockIdx.y;
    long long n_in = n_in_max;
    if (n_ptr) {
        const long long m = n_ptr[(long long)s * n_ptr_stride];
        n_in = m < n_in ? (m < 0 ? 0 : m) : n_in;
    }
    const long long n = n_in * k;
    const unsigned char *__restrict__ x = in + (long long)s * in_stride;
    unsigned char *__restrict__ y = out + (long long)s * out_stride;
    if (n_out && blockIdx.x == 0 && threadIdx.x == 0) n_out[(long long)s * n_out_stride] = (int)n;
    if (k == 2 && (((uintptr_t)x) & 7) == 0 && (((uintptr_t)y) & 15) == 0) {
        // dibits: 8 symbols (one 8-byte load) -> 16 bits (one 16-byte store) per lane and trip (it was a byte store per bit)
        const long long n8 = n_in >> 3;
        for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (long long)gridDim.x * blockDim.x) {
            const uint2 v = reinterpret_cast<const uint2 *>(x)[c];
            // byte b of a word -> bytes (2b, 2b+1) = (bit 1, bit 0) of the symbol
            auto spread = [](unsigned w2) -> unsigned {      // two symbols (low 16 bits: s0 | s1 << 8) -> four output bytes
                const unsigned s0 = w2 & 0xffu, s1 = (w2 >> 8) & 0xffu;
                return ((s0 >> 1) & 1u) | ((s0 & 1u) << 8) | (((s1 >> 1) & 1u) << 16) | ((s1 & 1u) << 24);
            };
            uint4 o;
            o.x = spread(v.x); o.y = spread(v.x >> 16); o.z = spread(v.y); o.w = spread(v.y >> 16);
            reinterpret_cast<uint4 *>(y)[c] = o;
        }
        for (long long i = 16 * n8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const long long q = i / k;
            const unsigned j = k - 1 - (unsigned)(i - q * k);
            y[i] = (unsigned char)(((unsigned)x[q] >> j) & 1u);
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long q = i / k;
        const unsigned j = k - 1 - (unsigned)(i - q * k);
        y[i] = (unsigned char)(((unsigned)x[q] >> j) & 1u);
    }
}

int launch_unpack_k_bits_streams(unsigned k, int n_streams, const unsigned char *in, long long in_stride, unsigned char *out,
                                 long long out_stride, long long n_in_max, const int *n_ptr, int n_ptr_stride, int *n_out,
                                 int n_out_stride, hipStream_t st)
{
    if (n_streams <= 0 || n_in_max <= 0) return GRHIP_OK;
    long long blocks = (n_in_max * k + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(unpack_k_bits_streams_kernel, dim3((unsigned)blocks, (unsigned)n_streams), dim3(256), 0, st, k, in, in_stride, out,
                       out_stride, n_in_max, n_ptr, n_ptr_stride, n_out, n_out_stride);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

int launch_unpack_k_bits(unsigned k, const unsigned char *in, unsigned char *out, long long noutput_items, hipStream_t st)
{
    if (noutput_items <= 0) return GRHIP_OK;
    long long blocks = (noutput_items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(unpack_k_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, k, in, out, noutput_items);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// binary slicer (gr-digital/lib/digital_binary_slicer_fb.cc:54-56)
// ===========================================================================
// 16 items per lane and trip: four 16-byte loads in flight, one 16-byte store (5 B of traffic per item)
__global__ void __launch_bounds__(256)
slicer_kernel(const float *__restrict__ in, unsigned char *__restrict__ out, long long n, int vec)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (vec) {
        const long long n16 = n >> 4;
        for (long long c = tid; c < n16; c += stride) {
            const float4 *src = reinterpret_cast<const float4 *>(in) + 4 * c;
            const float4 a = src[0], b = src[1], d = src[2], e = src[3];
            uint4 o;
            o.x = (a.x >= 0) | ((a.y >= 0) << 8) | ((a.z >= 0) << 16) | ((a.w >= 0) << 24);
            o.y = (b.x >= 0) | ((b.y >= 0) << 8) | ((b.z >= 0) << 16) | ((b.w >= 0) << 24);
            o.z = (d.x >= 0) | ((d.y >= 0) << 8) | ((d.z >= 0) << 16) | ((d.w >= 0) << 24);
            o.w = (e.x >= 0) | ((e.y >= 0) << 8) | ((e.z >= 0) << 16) | ((e.w >= 0) << 24);
            reinterpret_cast<uint4 *>(out)[c] = o;
        }
        for (long long i = (n16 << 4) + tid; i < n; i += stride) out[i] = (in[i] >= 0) ? 1 : 0;
    } else {
        for (long long i = tid; i < n; i += stride) out[i] = (in[i] >= 0) ? 1 : 0;
    }
}

int launch_binary_slicer(const float *in, unsigned char *out, long long n, hipStream_t st)
{
    if (n <= 0) return GRHIP_OK;
    const int vec = ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
    long long blocks = ((vec ? n / 16 + 1 : n) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(slicer_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n, vec);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// correlate_access_code_bb (gr-digital/lib/digital_correlate_access_code_bb.cc:95-130)
// Closed form of the shift-register loop: with registers
// (data_reg, flag_reg) at the start of the call,
//   out[i] bit0 = bit i-64 of the input stream            (data_reg for i < 64)
//   out[i] bit1 = flag_reg bit (63-i) for i < 64,  OR
//                 [popcount((W_k ^ code) & mask) <= thr],  k = i - len >= 0
//   W_k = the 64 stream bits before position k, oldest at the MSB.
// A 256-lane workgroup owns CORR_TB outputs: it packs the input bits it needs
// into 64-bit words in LDS with wave ballots (coalesced byte / float loads),
// then every lane emits 8 output bytes with one 8-byte store.  HBM-bound:
// 1 B (or 4 B when the slicer is fused) in + 1 B out per item.
// ===========================================================================
constexpr int CORR_TB = 8192;     // outputs per workgroup (4 rounds of 8 per lane)
constexpr int CORR_WORDS = CORR_TB / 64 + 2;

#ifndef GRHIP_CORR_TPW
#define GRHIP_CORR_TPW 4
#endif
constexpr int CORR_TPW = GRHIP_CORR_TPW;        // consecutive tiles per workgroup

// One workgroup walks CORR_TPW consecutive tiles of a stream; the loads of the next tile (interior tiles: 16-byte loads
// kept in registers) are issued before the current tile is packed and evaluated, so HBM latency runs under the bit work.
template <bool SOFT>        // SOFT: float items in (the slicer fused), else bytes
__global__ void __launch_bounds__(256)
corr_kernel(CorrParams p, const CorrState *__restrict__ state_in, const unsigned char *__restrict__ in_bytes,
            const float *__restrict__ in_soft, long long in_stride, unsigned char *__restrict__ out,
            long long out_stride, long long n_arg, const int *__restrict__ n_ptr, int n_ptr_stride)
{
    __shared__ unsigned long long P[CORR_WORDS + 1];    // + one word of slack for the unconditional look-ahead read
    const int s = blockIdx.y, t = threadIdx.x;
    long long n = n_arg;
    if (n_ptr) { long long m = n_ptr[(long long)s * n_ptr_stride]; n = m < n ? m : n; }
    if ((long long)blockIdx.x * CORR_TPW * CORR_TB >= n) return;
    const unsigned char *__restrict__ xb = !SOFT ? in_bytes + (long long)s * in_stride : nullptr;
    const float *__restrict__ xf = SOFT ? in_soft + (long long)s * in_stride : nullptr;
    const CorrState st = state_in[s];
    if (t == 0) P[CORR_WORDS] = 0;      // slack word (read by the look-ahead, its bits are never used)
    const unsigned chi = (unsigned)(p.access_code >> 32), clo = (unsigned)p.access_code;
    const unsigned mhi = (unsigned)(p.mask >> 32), mlo = (unsigned)p.mask;
    unsigned char *__restrict__ y = out + (long long)s * out_stride;

    // words i0/64 - 2 .. i0/64 + CORR_TB/64 - 1 of a tile, first item of a word at its MSB.
    // Wide path (interior tiles, 16-byte aligned): a lane takes 16 input bytes (or 8 floats) with 16-byte loads, squeezes
    // their LSBs (sign decisions) into one 16-bit (8-bit) chunk with a multiply, and drops the chunk into its place inside
    // the 64-bit word (little-endian LDS: chunk c of a word lives at sub-index last-c).
    constexpr int NCHB = CORR_WORDS * 4, PERB = (NCHB + 255) / 256;       // 16-item chunks (bytes)
    constexpr int NCHF = CORR_WORDS * 8, PERF = (NCHF + 255) / 256;       // 8-item chunks (floats)
    uint4 vb[SOFT ? 1 : PERB];
    float4 vf[SOFT ? 2 * PERF : 1];
    auto is_wide = [&](long long i0) -> bool {
        const long long base = (i0 / 64 - 2) * 64;
        return (base >= 0) && (base + (long long)CORR_WORDS * 64 <= n) &&
               ((((uintptr_t)(xb ? (const void *)(xb + base) : (const void *)(xf + base))) & 15) == 0);
    };
    auto request = [&](long long i0) __attribute__((always_inline)) {
        const long long base = (i0 / 64 - 2) * 64;
        if (!SOFT) {
#pragma unroll
            for (int k = 0; k < PERB; ++k) {
                const int c = t + 256 * k;
                vb[k] = (c < NCHB) ? reinterpret_cast<const uint4 *>(xb + base)[c] : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < PERF; ++k) {
                const int c = t + 256 * k;
                const float4 *src = reinterpret_cast<const float4 *>(xf + base) + 2 * (c < NCHF ? c : 0);
                vf[2 * k] = src[0];
                vf[2 * k + 1] = src[1];
            }
        }
    };
    // (the grid may be smaller than the stream -- the launcher sizes it by the expected item count: a workgroup goes on
    // in strides of the grid)
    for (long long tile0 = (long long)blockIdx.x * CORR_TPW; tile0 * CORR_TB < n; tile0 += (long long)gridDim.x * CORR_TPW) {
    bool wide = is_wide(tile0 * CORR_TB);
    if (wide) request(tile0 * CORR_TB);

    for (int ti = 0; ti < CORR_TPW; ++ti) {
        const long long i0 = (tile0 + ti) * CORR_TB;
        if (i0 >= n) break;
        const long long wlo = i0 / 64 - 2;
        if (wide && !SOFT) {
            unsigned short *P16 = reinterpret_cast<unsigned short *>(P);
#pragma unroll
            for (int k = 0; k < PERB; ++k) {
                const int c = t + 256 * k;
                if (c < NCHB) {
                    // ((w & 0x01010101) * 0x08040201) >> 24 : LSBs of bytes 0..3 -> bits 3..0
                    const unsigned n0 = (((vb[k].x & 0x01010101u) * 0x08040201u) >> 24) & 0xFu;
                    const unsigned n1 = (((vb[k].y & 0x01010101u) * 0x08040201u) >> 24) & 0xFu;
                    const unsigned n2 = (((vb[k].z & 0x01010101u) * 0x08040201u) >> 24) & 0xFu;
                    const unsigned n3 = (((vb[k].w & 0x01010101u) * 0x08040201u) >> 24) & 0xFu;
                    P16[(c & ~3) + (3 - (c & 3))] = (unsigned short)((n0 << 12) | (n1 << 8) | (n2 << 4) | n3);
                }
            }
        } else if (wide && SOFT) {
            unsigned char *P8 = reinterpret_cast<unsigned char *>(P);
#pragma unroll
            for (int k = 0; k < PERF; ++k) {
                const int c = t + 256 * k;
                if (c < NCHF) {
                    const float4 a0 = vf[2 * k], a1 = vf[2 * k + 1];
                    const unsigned b = ((a0.x >= 0) << 7) | ((a0.y >= 0) << 6) | ((a0.z >= 0) << 5) | ((a0.w >= 0) << 4) |
                                       ((a1.x >= 0) << 3) | ((a1.y >= 0) << 2) | ((a1.z >= 0) << 1) | (a1.w >= 0);
                    P8[(c & ~7) + (7 - (c & 7))] = (unsigned char)b;
                }
            }
        } else {
            // first / last tile of a stream, or unaligned input: one item per lane, wave ballot
            for (int wi = t >> 6; wi < CORR_WORDS; wi += 4) {
                long long w = wlo + wi;
                long long idx = w * 64 + (t & 63);
                int bit = 0;
                if (idx >= 0 && idx < n) bit = xb ? (xb[idx] & 1) : (xf[idx] >= 0 ? 1 : 0);
                unsigned long long m = __ballot(bit);            // lane l -> bit l
                if ((t & 63) == 0) P[wi] = __brevll(m);          // first item at the MSB
            }
        }
        // the next tile's loads fly under this tile's bit work
        {
            const long long i1 = i0 + CORR_TB;
            wide = ti + 1 < CORR_TPW && i1 < n && is_wide(i1);
            if (wide) request(i1);
        }
        __syncthreads();

        // words before the call: w = -1 is the shift register carried in, older ones are zero
        // (the ballot path above has packed zeros there; only the first tile of a stream has them)
        if (i0 == 0) {
            if (t == 0) P[1] = st.data_reg;
            __syncthreads();
        }

        // A lane emits 8 consecutive outputs per round.  Their bit 0 is one byte of one packed
        // word; their flags come from ONE 64-bit window that slides by a bit per output
        // (two 32-bit halves: a funnel shift and a shift-or), instead of re-assembling the
        // window from LDS for every output.
        for (int rnd = 0; rnd < CORR_TB / 2048; ++rnd) {
            const long long o0 = i0 + 2048ll * rnd + 8 * t;
            if (o0 >= n) break;
            // bit 0 of outputs o0 .. o0+7: stream bits o0-64 .. o0-57 (same word, byte aligned)
            const long long b0 = o0 - 64;
            const unsigned byte0 = (unsigned)(P[(b0 >> 6) - wlo] >> (56 - (int)(b0 & 63))) & 0xFFu;
            // window of output o0: stream bits k0-64 .. k0-1, k0 = o0 - len; then bits k0 .. k0+6 slide in
            const long long k0 = o0 - (long long)p.len;
            const long long wk = k0 >> 6;                          // floor division (arithmetic shift)
            const int rk = (int)(k0 & 63);
            const unsigned long long hi = P[wk - 1 - wlo], lo = P[wk - wlo], nx = P[wk + 1 - wlo];
            const unsigned long long W0 = rk ? (hi << rk) | (lo >> (64 - rk)) : hi;
            const unsigned nb = (unsigned)((rk ? (lo << rk) | (nx >> (64 - rk)) : lo) >> 56);
            // The eight windows are eight funnel shifts of (W0, next byte) by constant amounts -- no chain from one output to
            // the next; a flag is a carry into an 8-bit accumulator (acc + acc + [nwrong <= thr]: bit 7-j = output j, the
            // layout of byte0); both bytes are then spread to one bit per output byte with a multiply.
            const unsigned whi0 = (unsigned)(W0 >> 32), wlo0 = (unsigned)W0, nbtop = nb << 24;
            unsigned acc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned wh = j ? __builtin_amdgcn_alignbit(whi0, wlo0, 32 - j) : whi0;
                const unsigned wl = j ? __builtin_amdgcn_alignbit(wlo0, nbtop, 32 - j) : wlo0;
                const unsigned nwrong = (unsigned)__popc((wh ^ chi) & mhi) + (unsigned)__popc((wl ^ clo) & mlo);
                acc = acc + acc + (nwrong <= p.threshold ? 1u : 0u);
            }
            // positions before the stream (k0 + j < 0) raise no flag; nor does an empty code
            const unsigned valid = (p.len == 0 || k0 <= -8) ? 0u : (k0 >= 0 ? 0xFFu : (0xFFu >> (unsigned)(-k0)));
            acc &= valid;
            if (o0 < 64)            // flags already in flight at call start: flag_reg bit 63 - (o0 + j) for output j
                acc |= (unsigned)((st.flag_reg >> (56 - (int)o0)) & 0xFFull);
            // four bits b3 b2 b1 b0 -> bytes (b3, b2, b1, b0) from the low byte up: the partial products of the multiply
            // occupy disjoint bit ranges (no carries)
            // (24-bit multiply: full rate; the fourth partial product is a shift)
            auto spread4 = [](unsigned x4) { return (((__umul24(x4, 0x8040u)) | (x4 << 24)) & 0x01010100u) | (x4 >> 3); };
            const unsigned out_lo = spread4(byte0 >> 4) | (spread4(acc >> 4) << 1);
            const unsigned out_hi = spread4(byte0 & 15u) | (spread4(acc & 15u) << 1);
            if (o0 + 8 <= n && ((((uintptr_t)(y + o0)) & 7) == 0)) {
                *reinterpret_cast<uint2 *>(y + o0) = make_uint2(out_lo, out_hi);
            } else {
                const unsigned long long packed = ((unsigned long long)out_hi << 32) | out_lo;
                for (int j = 0; j < 8; ++j)
                    if (o0 + j < n) y[o0 + j] = (unsigned char)(packed >> (8 * j));
            }
        }
        __syncthreads();            // P belongs to the next tile from here
    }
    }
}

// registers after n items (one 64-lane workgroup per stream)
__global__ void __launch_bounds__(64)
corr_tail_kernel(CorrParams p, CorrState *__restrict__ state, const unsigned char *__restrict__ in_bytes,
                 const float *__restrict__ in_soft, long long in_stride, long long n_arg,
                 const int *__restrict__ n_ptr, int n_ptr_stride)
{
    const int s = blockIdx.x, t = threadIdx.x;
    long long n = n_arg;
    if (n_ptr) { long long m = n_ptr[(long long)s * n_ptr_stride]; n = m < n ? m : n; }
    if (n <= 0) return;
    const unsigned char *__restrict__ xb = in_bytes ? in_bytes + (long long)s * in_stride : nullptr;
    const float *__restrict__ xf = in_soft ? in_soft + (long long)s * in_stride : nullptr;
    const CorrState st = state[s];
    auto bit_at = [&](long long idx) -> unsigned long long {
        if (idx >= n) return 0ull;
        if (idx >= 0) return xb ? (unsigned long long)(xb[idx] & 1) : (xf[idx] >= 0 ? 1ull : 0ull);
        long long back = -1 - idx;                  // in[-1] is data_reg bit 0
        return back < 64 ? (st.data_reg >> back) & 1ull : 0ull;
    };
    // The last 128 stream bits, two wave ballots (every lane loads two items, all in flight at once; a serial walk of 64
    // dependent loads per lane cost 17 us per call): A = bits n-128 .. n-65, B = bits n-64 .. n-1, oldest at the MSB.
    const unsigned long long A = __brevll(__ballot((int)bit_at(n - 128 + t)));
    const unsigned long long B = __brevll(__ballot((int)bit_at(n - 64 + t)));
    // lane t evaluates the flag of position k = n - 1 - t (for t < len), which sits at
    // flag_reg bit (64 - len) + t after n items.  Its window, bits k-64 .. k-1, is (A:B) >> (t + 1).
    unsigned long long contrib = 0;
    if (t < (int)p.len) {
        long long k = n - 1 - t;
        if (k >= 0) {
            const unsigned long long W = t == 63 ? A : (A << (63 - t)) | (B >> (t + 1));
            unsigned nwrong = (unsigned)__popcll((W ^ p.access_code) & p.mask);
            if (nwrong <= p.threshold) contrib = 1ull << (64 - p.len + t);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        unsigned lo = (unsigned)contrib, hi = (unsigned)(contrib >> 32);
        lo |= __shfl_xor(lo, o); hi |= __shfl_xor(hi, o);
        contrib = ((unsigned long long)hi << 32) | lo;
    }
    if (t == 0) {
        unsigned long long f = (n < 64) ? (st.flag_reg << n) : 0ull;
        CorrState ns;
        ns.data_reg = B;
        ns.flag_reg = f | contrib;
        state[s] = ns;
    }
}

int launch_correlate(const CorrParams &p, CorrState *state, int n_streams, const unsigned char *in_bytes,
                     const float *in_soft, long long in_stride, unsigned char *out, long long out_stride,
                     long long n, const int *n_ptr, int n_ptr_stride, hipStream_t st, long long n_expect)
{
    if (n <= 0 || n_streams <= 0) return GRHIP_OK;
    const long long per_wg = (long long)CORR_TB * CORR_TPW;
    // n is a capacity when n_ptr gives the streams' item counts: the grid covers what is expected (workgroups past a
    // stream's count leave at once, but a million of them cost milliseconds), longer streams are walked in strides
    const long long n_grid = n_expect > 0 && n_expect < n ? n_expect : n;
    dim3 grid((unsigned)((n_grid + per_wg - 1) / per_wg), (unsigned)n_streams);
    if (in_bytes) hipLaunchKernelGGL(corr_kernel<false>, grid, dim3(256), 0, st, p, (const CorrState *)state, in_bytes, in_soft,
                                     in_stride, out, out_stride, n, n_ptr, n_ptr_stride);
    else hipLaunchKernelGGL(corr_kernel<true>, grid, dim3(256), 0, st, p, (const CorrState *)state, in_bytes, in_soft,
                            in_stride, out, out_stride, n, n_ptr, n_ptr_stride);
    GRHIP_HIP(hipGetLastError());
    hipLaunchKernelGGL(corr_tail_kernel, dim3(n_streams), dim3(64), 0, st, p, state, in_bytes, in_soft,
                       in_stride, n, n_ptr, n_ptr_stride);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

// ===========================================================================
// gr_stream_to_streams / gr_streams_to_stream (general/gr_stream_to_streams.cc:46-66,
// general/gr_streams_to_stream.cc:46-69): item i of stream j <-> item i*nstreams + j of the
// single stream.  One lane per i: it touches nstreams consecutive items of the single stream
// (coalesced across the wave) and one item of each separate stream (coalesced per stream).
// Items are copied as 8-byte words when their size allows it, else 4-byte or single bytes.
// ===========================================================================
template <class W, bool SPLIT>
__global__ void __launch_bounds__(256)
streams_kernel(W *__restrict__ single, W *__restrict__ multi, long long multi_stride_w, int nstreams, int wpi, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        for (int j = 0; j < nstreams; ++j) {
            W *a = single + (i * nstreams + j) * wpi;
            W *b = multi + (long long)j * multi_stride_w + i * wpi;
            for (int w = 0; w < wpi; ++w) {
                if (SPLIT) b[w] = a[w];
                else a[w] = b[w];
            }
        }
    }
}

int launch_streams(bool split, void *single, void *multi, long long multi_stride_items, int nstreams, size_t item_size,
                   long long n_items_per_stream, hipStream_t st)
{
    if (n_items_per_stream <= 0 || nstreams <= 0) return GRHIP_OK;
    long long blocks = (n_items_per_stream + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const uintptr_t al = (uintptr_t)single | (uintptr_t)multi | (uintptr_t)(multi_stride_items * (long long)item_size);
#define GRHIP_STREAMS(W)                                                                                         \
    do {                                                                                                       \
        const int wpi = (int)(item_size / sizeof(W));                                                          \
        const long long msw = multi_stride_items * wpi;                                                        \
        if (split) hipLaunchKernelGGL((streams_kernel<W, true>), dim3((unsigned)blocks), dim3(256), 0, st, (W *)single, \
                                      (W *)multi, msw, nstreams, wpi, n_items_per_stream);                    \
        else hipLaunchKernelGGL((streams_kernel<W, false>), dim3((unsigned)blocks), dim3(256), 0, st, (W *)single,      \
                                (W *)multi, msw, nstreams, wpi, n_items_per_stream);                          \
    } while (0)
    if (item_size % 8 == 0 && (al & 7) == 0) GRHIP_STREAMS(unsigned long long);
    else if (item_size % 4 == 0 && (al & 3) == 0) GRHIP_STREAMS(unsigned int);
    else GRHIP_STREAMS(unsigned char);
#undef GRHIP_STREAMS
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

}  // namespace grhip
