// fir_tiled.hip -- the throughput kernel of the FIR family (complex data):
// gr_fir_ccf / gr_fir_ccc decimating FIR, the composite FIR + rotator of
// gr_freq_xlating_fir_filter_ccc, the fused quadrature demodulator, and (float-pair
// mode) gr_fir_fff.
//
// Shape of the work: y[n] = sum_k c[k] x[nD+k] is a vector x scalar recurrence
// (wave-uniform taps against per-lane data), VALU-bound at 256 taps,
// so the design goal is to keep the vector FMA pipes issuing while HBM traffic,
// LDS staging and the epilogue hide underneath.  No MFMA.
//
//  * Persistent 256-lane workgroups (2 per CU; 4 for plain decimation-1 filters) walk
//    tiles of NT = 256*R outputs over all streams of the launch; the first two tiles of
//    a workgroup are static, the rest come from a tile queue (one atomic per tile,
//    requested two tiles ahead).
//  * The input tile is fetched from HBM with 16-byte raw buffer loads into REGISTERS
//    one tile ahead (the hardware range check supplies the history zeros and the end
//    of the stream: no bounds code, no branch around a load), a share of the loads at
//    the top of each polyphase pass of the MAC loop.
//  * The tile is then written to LDS de-interleaved into its D polyphase
//    components (x_p[m] = x[mD+p]) with one pad slot per R samples, so that the
//    lane stride is R+1 (odd) 8-byte slots: conflict-free ds_read_b64.  The pad slots
//    of component 0 hold the demodulator's arctangent table.
//  * Each lane keeps R complex accumulators.  Its samples come in blocks of R; a step
//    multiplies R taps against two blocks (R*R v_pk_fma_f32, tap = SGPR operand).  For
//    real taps a step is ONE asm statement: the scalar load of the next step's taps,
//    the FMAs, the wait -- see the MAC loop.
//  * Real prototype taps (the usual low-pass) take the pre-mix form of the
//    frequency translation: x'[u] = x[u] * e^{jw(u-D)} at staging (phasors kept in
//    VGPRs), real-tap MACs: half the flops of complex taps.
//  * Epilogues: EPI_ROTATE = per-output phase correction of the pre-mix form + rotator
//    multiply with the exact-recurrence phase table (8 B / output); EPI_DEMOD = the
//    fused quadrature demodulator working directly on the pre-mixed accumulators, its
//    y[n-1] made local by overlapping one lane per wave.
#include <cstdio>
#include <cstdlib>

#include "device_math.h"
#include "fir_kernels.h"
#include "grhip_internal.h"

namespace grhip {

// R = outputs per lane.  R = 8: one LDS read per 8 packed FMAs, 2 workgroups (8 waves)
// per CU.  R = 4: twice the LDS reads per FMA but 3 workgroups (12 waves) per CU.
#ifndef GRHIP_TILED_R
#define GRHIP_TILED_R 8
#endif
constexpr int TILED_R = GRHIP_TILED_R;
constexpr int TILED_LOGR = TILED_R == 8 ? 3 : 2;
static_assert(TILED_R == 8 || TILED_R == 4, "R must be 4 or 8");
#ifndef GRHIP_TILED_THREADS
#define GRHIP_TILED_THREADS 256
#endif
constexpr int TILED_THREADS = GRHIP_TILED_THREADS;
constexpr int TILED_NT = TILED_THREADS * TILED_R;
constexpr int TILED_WG_PER_CU = (TILED_R == 8 ? 2 : 3) * 256 / TILED_THREADS;
constexpr int TILED_NI = TILED_R == 8 ? 18 : 11;       // 16-byte loads per lane per tile at decimation 4 (upper bound)
// rounds of loads a tile can need at decimation D: (NT + Tq) D / 512 with up to 1024 taps
// plain decimation-1 filters stage 6 rounds instead of 18: their registers fit four workgroups
// per CU (the LDS tile of a short filter is small), which hides more of the per-tile latencies
__host__ __device__ constexpr int tiled_wg_per_cu(int D, bool premix, int epi)
{
    return (D == 1 && !premix && epi == 0) ? 2 * TILED_WG_PER_CU : TILED_WG_PER_CU;
}
__host__ __device__ constexpr int tiled_ni(int D) { return D >= 4 ? TILED_NI : D == 2 ? 10 : 6; }
constexpr int TILED_LDS_LIMIT = (160 * 1024) / TILED_WG_PER_CU - 256;

int tiled_R() { return TILED_R; }
int tiled_NT() { return TILED_NT; }
int tiled_wtab_len() { return 2 * TILED_THREADS + 1; }
int tiled_load_span() { return 2 * TILED_THREADS; }
int tiled_stab_len() { return TILED_NI; }

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// read-only, wave-uniform operands: constant address space => s_load
typedef const float __attribute__((address_space(4))) *cfloat_p;

// LDS geometry (float2 slots).  mm = m + R where m is the polyphase sample index
// relative to the tile's first real output (m = -1: first sample of the boundary
// output).  slot(mm) = mm + mm/R.
__host__ __device__ inline int tiled_phase_stride(int Tq)
{
    int MM = TILED_NT + Tq + 2 * TILED_R;   // largest mm touched: NT + Tq + 2R - 1 (MAC loop look-ahead reads)
    return MM + (MM >> TILED_LOGR) + 1;
}
__host__ inline size_t tiled_lds_bytes(int D, int Tq)
{
    return (size_t)D * tiled_phase_stride(Tq) * sizeof(float2) + 16;      // + the scheduler's hand-over slot
}

// The arctangent table of the fused demodulator lives in the pad slots of polyphase
// component 0 (slot 9k+8 for R = 8, never touched by staging or the MAC loop): pad k holds
// the pair (tab[k], tab[k+1]), so the two entries an interpolation needs come with one
// 8-byte read and the table takes no LDS of its own.
struct PadPairs {
    float __attribute__((ext_vector_type(2))) *xs2;
    __device__ __forceinline__ float __attribute__((ext_vector_type(2))) &operator[](int k) const
    {
        return xs2[(k << TILED_LOGR) + k + TILED_R];
    }
};
static_assert(((TILED_NT + 2 * TILED_R + TILED_R) >> TILED_LOGR) >= 256, "pad slots must hold the arctangent table");

bool tiled_supported(int decim, int Tq)
{
    if (!(decim == 1 || decim == 2 || decim == 4)) return false;
    if (Tq <= 0 || (Tq % TILED_R) != 0) return false;
    if ((TILED_NT + Tq) * decim > tiled_ni(decim) * 2 * TILED_THREADS) return false;
    return tiled_lds_bytes(decim, Tq) <= (size_t)TILED_LDS_LIMIT;
}

// Diagnostic build only (-DGRHIP_STAMP, `make stamp`): per-wave cycle shares of the
// phases of a tile, written to a buffer of their own (never to an output).
#ifdef GRHIP_STAMP
__device__ unsigned long long *g_stamp_buf = nullptr;
#define STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = stamp_now(), st_begin = st_last
#define STAMP(k) do { unsigned long long n__ = stamp_now(); st_acc[k] += n__ - st_last; st_last = n__; } while (0)
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");   // constant 100 MHz
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#else
#define STAMP_DECL
#define STAMP(k)
#endif

// sum over the 64 lanes of a wave with DPP adds (no LDS traffic); every lane gets the
// total.  quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror give each row of
// 16 its sum; row_bcast15 / row_bcast31 carry it across rows into lane 63.
__device__ __forceinline__ float wave_sum(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xc, 0xf, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// FP > 0: float-pair mode for gr_fir_fff.  A float FIR with decimation FP is the
// complex-data kernel with decimation D = 2 FP run on the overlapped pair stream
// z[m] = (x[m], x[m+FP]):  sum_k c[k] z[nD + k] = (y[2n], y[2n+1]).  The pair stream is
// never materialised: each lane loads 16 bytes (4 floats) at an 8-byte lane stride and
// forms its two items while staging; the outputs land in a plain float array.
template <int D, bool CTAPS, bool PREMIX, int EPI, int FP = 0>
__global__ void __launch_bounds__(TILED_THREADS, tiled_wg_per_cu(D, PREMIX, EPI)) fir_tiled_kernel(const FirTiledArgs a)
{
    static_assert(FP == 0 || (D == 2 * FP && !CTAPS && !PREMIX && EPI == EPI_NONE), "float-pair mode");
    constexpr int R = TILED_R, LOGR = TILED_LOGR, NT = TILED_NT, NI = tiled_ni(D);
    constexpr int LOGD = ilog2(D);
    constexpr int TW = CTAPS ? 2 : 1;               // floats per tap
    constexpr bool ROT = EPI == EPI_ROTATE || EPI == EPI_ROTATE_DEMOD;     // rotator table multiply
    constexpr bool DEMOD = EPI == EPI_ROTATE_DEMOD || EPI == EPI_DEMOD;    // fused quadrature demodulator
    constexpr bool DIRECT = EPI == EPI_DEMOD;                              // ... on the pre-mixed accumulators
    static_assert(!DIRECT || PREMIX, "EPI_DEMOD is the pre-mix form's epilogue");
    static_assert(EPI != EPI_ROTATE_DEMOD, "the only fused demodulator is EPI_DEMOD");
    // The demodulator needs y[n-1] next to y[n].  Inside a wave that is the neighbouring
    // lane's last accumulator (one DPP shift); across waves and tiles it is made local by
    // OVERLAP: lane 0 of every wave recomputes the R outputs of the lane before it (the
    // previous wave's or the previous tile's last lane) and stores nothing.  A tile
    // therefore covers NTC = NT - 3R distinct outputs, NTE = NTC - R of them new: 1.6 %
    // more MACs instead of a reduction over the taps, its LDS reads and its latency.
    constexpr int WAVES = TILED_THREADS / 64;
    constexpr int OVL = DIRECT ? TILED_R : 0;
    constexpr int NTC = TILED_NT - OVL * (WAVES - 1);
    constexpr int NTE = NTC - OVL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *xs = (float2 *)smem;

    const int t = threadIdx.x;
    const int Tq = a.Tq;
    const int PS = tiled_phase_stride(Tq);
    const int Lu = (NTC + Tq) * D;                  // samples per tile, u = 0 <-> (first covered output - 1)*D
    const int tl = t - (OVL ? (t >> 6) : 0);        // lane's place in the tile: outputs tl*R .. tl*R+R-1 of the covered range
    // tiles are numbered (stream, tile-in-stream); the pair is advanced incrementally
    // (a 64-bit division per tile costs ~150 scalar instructions)
    const int tiles_per_stream = (int)((a.n_out + NTE - 1) / NTE);
    const cfloat_p hp = (cfloat_p)a.hp;             // phase-major taps, padded by R entries

    // lane constants of the pre-mix phasor: W[v] = e^{jw(v-D)}, v = 2t-1, 2t, 2t+1
    // (table entry v+1 holds W[v], v = -1 .. 511)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 wA{1.f, 0.f}, wB = wA;
    if (PREMIX) {
        const f32x2 *wt = reinterpret_cast<const f32x2 *>(a.wtab);
        wA = wt[2 * t];
        wB = wt[2 * t + 1];
    }

    // per-lane output phase corrections of the pre-mix form (tile independent)
    float2 vreg[R];
    if (PREMIX && !DIRECT) {
        const float4 *vv = reinterpret_cast<const float4 *>(a.vtab + t * R);   // 64-byte aligned
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            const float4 v2 = vv[r >> 1];
            vreg[r] = make_float2(v2.x, v2.y);
            vreg[r + 1] = make_float2(v2.z, v2.w);
        }
    }
    // arctangent table of the fused demodulator lives in LDS (1 KB)
    const PadPairs s_atan{reinterpret_cast<float __attribute__((ext_vector_type(2))) *>(xs)};
    // Pre-mix phasor of the lane's first sample in every staging round, W[ub + 512 i] =
    // (lane constant) x (step e^{jw 512 i}): tile independent, built once, kept in VGPRs
    // (36 of them; the second sample's phasor is one more product with e^{jw}).  They depend
    // on the parity `off` of the stream's 16-byte alignment, which is the same for every
    // tile of a launch unless the stream stride is odd: rebuilt when it changes.
    f32x2 W0[NI];
    const f32x2 wstep = cmul_pk(wB, f32x2{wA.x, -wA.y});      // e^{jw}
    int w_off = -1;
    if (DEMOD) {
        for (int i = t; i < 256; i += TILED_THREADS)
            s_atan[i] = (float __attribute__((ext_vector_type(2)))){a.atan_tab[i], a.atan_tab[i + 1]};
    }

    float4 pf[NI];
    float2 gq[R];                          // rotator phases of the lane's outputs

    // tile -> buffer descriptor of its stream + byte offset of the lane's first pair.
    // The stream is addressed through a raw buffer descriptor whose base is the first
    // real item (index n_lo) and whose size is the readable part: lanes that fall
    // before the stream (the history zeros of a fresh flowgraph) or past its end get
    // zeros from the hardware range check (per dword), so there is no bounds code.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int LANE_BYTES = FP ? 8 : 16;         // bytes between the loads of neighbouring lanes
    int lead = 0;       // set by tile_geom
    auto tile_geom = [&](int s, int b, __amdgpu_buffer_rsrc_t &rsrc, int &voff, int &off) {
        const long long g0 = ((long long)b * NTE - OVL - 1) * D - a.n_lo;   // tile start relative to the first real item
        if (FP) {
            // items are floats; strides and counts of the launch are in floats
            const float *xf = reinterpret_cast<const float *>(a.x) + (long long)s * a.x_stride + a.n_lo;
            off = 0; lead = 0;
            const long long bytes = (a.n_in - a.n_lo) * 4;
            rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xf), 0, (int)bytes, 0x00020000);
            voff = (int)(g0 * 4) + LANE_BYTES * t;
            return;
        }
        const float2 *x = a.x + (long long)s * a.x_stride + a.n_lo;       // first real item
        const long long unit0 = (long long)(((unsigned long long)(uintptr_t)x) >> 3) + g0;
        off = (int)(unit0 & 1);                     // pair starts on a 16-byte boundary
        // The descriptor must start on a 16-byte boundary too: a 16-byte load at offset -8 is out of range as a
        // WHOLE (no per-dword wrap-around), which would lose the stream's first item.  When that item sits on an
        // odd 8-byte unit (lead = 1) the descriptor starts one item earlier -- the same 16-byte granule, hence the
        // same allocation -- and stage() zeroes that item where a window can see it (round 2 fix: a stream that
        // started 8 bytes off a 16-byte boundary lost its first item; no caller of round 1 produced one).
        lead = (int)((((unsigned long long)(uintptr_t)x) >> 3) & 1);
        const long long bytes = (a.n_in - a.n_lo + lead) * 8;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(x - lead), 0, (int)bytes, 0x00020000);
        voff = (int)((g0 - off + lead) * 8) + LANE_BYTES * t;      // may be negative: out of range => zeros
    };

    // ---- rotator phases of a tile's outputs (issued at the end of the previous tile:
    //      a full stage + MAC phase ahead of their use, single register set) ----------
    auto fetch_phases = [&](int b) {
        if (ROT) {
            // uniform base + 32-bit lane offsets; indices clamped to the last valid output
            const float2 *gt = a.gtab + (long long)b * NT;
            const long long left = a.n_out - (long long)b * NT;
            const int lim = (int)(left < NT ? left : NT) - 1;          // last valid tile-local index
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int j = t * R + r;
                gq[r] = gt[j < lim ? j : lim];
            }
        }
    };

    // ---- issue the HBM loads of one tile into registers ------------------------------
    // part < 0: all NI loads; otherwise the part-th of D equal shares (the shares are
    // issued at the top of the D polyphase passes of the MAC loop: a burst of NI
    // 1 KB loads per wave overruns the CU's memory queue and the wave sits at issue)
    auto fetch = [&](__amdgpu_buffer_rsrc_t rsrc, int voff, int part) {
        constexpr int PER = (NI + D - 1) / D;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (part >= 0 && i / PER != part) continue;
            // rounds past the end of the tile (short filters, small decimation) are pointed
            // out of range: zeros, no memory traffic, still no branch around a load
            constexpr int INSIDE = NTC * D / (2 * TILED_THREADS);      // rounds that every tile needs in full
            const int vo = (i < INSIDE || i * 2 * TILED_THREADS - 1 < Lu) ? voff + i * (LANE_BYTES * TILED_THREADS) : 0x7ffff000;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0);
            const f32x4 f = __builtin_bit_cast(f32x4, v);
            pf[i] = make_float4(f[0], f[1], f[2], f[3]);
        }
    };

    // ---- registers -> LDS, de-interleaved, pre-mixed ----------------------------
    // u = ub + 512 i  =>  m and the padded slot advance by constants per i:
    // slot_i = slot_0 + (512/D)(1 + 1/R) i, so the LDS addresses are immediates.
    auto stage = [&](int s, int b) {
        __amdgpu_buffer_rsrc_t rsrc; int voff, off;
        tile_geom(s, b, rsrc, voff, off);
        if (!FP && lead && b == 0 && a.n_lo > 0) {
            // the item in front of a stream that starts off a 16-byte boundary is a history zero
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (voff + i * (LANE_BYTES * TILED_THREADS) == 0) { pf[i].x = 0.f; pf[i].y = 0.f; }
        }
        if (PREMIX && off != w_off) {       // wave-uniform, first tile only in practice
            const f32x2 w0l = off ? wA : wB;
            const f32x2 *st = reinterpret_cast<const f32x2 *>(a.stab);
#pragma unroll
            for (int i = 0; i < NI; ++i) W0[i] = cmul_pk(w0l, st[i]);
            w_off = off;
        }
        const int ub = -off + 2 * t;
        constexpr int SLOT_STEP = (2 * TILED_THREADS / D) + (2 * TILED_THREADS / D) / R;
        constexpr int FULL = NTC * D / (2 * TILED_THREADS);    // rounds that lie inside the tile for every lane
        const int mm0 = (ub >> LOGD) - 1 + R, p0 = ub & (D - 1);
        const int mm1 = ((ub + 1) >> LOGD) - 1 + R, p1 = (ub + 1) & (D - 1);
        f32x2 *dst0 = reinterpret_cast<f32x2 *>(xs) + p0 * PS + mm0 + (mm0 >> LOGR);
        f32x2 *dst1 = reinterpret_cast<f32x2 *>(xs) + p1 * PS + mm1 + (mm1 >> LOGR);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u = ub + 2 * TILED_THREADS * i;
            f32x2 e0{pf[i].x, pf[i].y}, e1{pf[i].z, pf[i].w};
            if (FP == 1) { e0 = f32x2{pf[i].x, pf[i].y}; e1 = f32x2{pf[i].y, pf[i].z}; }   // (x[u],x[u+1]), (x[u+1],x[u+2])
            if (FP == 2) { e0 = f32x2{pf[i].x, pf[i].z}; e1 = f32x2{pf[i].y, pf[i].w}; }   // (x[u],x[u+2]), (x[u+1],x[u+3])
            if (PREMIX) {
                e0 = cmul_pk(e0, W0[i]);
                e1 = cmul_pk(e1, cmul_pk(W0[i], wstep));
            }
            if (i == 0) {
                if (u >= 0) dst0[0] = e0;
                dst1[0] = e1;
            } else if (i < FULL) {
                dst0[SLOT_STEP * i] = e0;
                dst1[SLOT_STEP * i] = e1;
            } else if (i * 2 * TILED_THREADS - 1 < Lu) {     // wave-uniform: the round exists
                if (u < Lu) dst0[SLOT_STEP * i] = e0;
                if (u + 1 < Lu) dst1[SLOT_STEP * i] = e1;
            }
        }
    };

    // tile id = b * n_streams + s (streams fastest): the workgroups that run at the same
    // time work on the same output range of different streams, so the shared rotator
    // phase table is served from L2 instead of being re-read from HBM once per stream.
    //
    // Scheduling: every workgroup takes tiles blockIdx and blockIdx + G statically, the rest
    // from a shared counter (a.sched).  The two workgroups of a CU do not run at the same
    // speed (the SIMDs favour the older wave: measured 10.1 vs 14.7 us per tile), so a static
    // split leaves the faster half idle for the last sixth of the launch.  The counter is
    // asked for tile i+2 while tile i is in its MAC loop, so its latency is never waited on.
    const unsigned total_tiles = (unsigned)tiles_per_stream * (unsigned)a.n_streams;
    const unsigned G = gridDim.x;
    unsigned cur = blockIdx.x, nxt = cur + G;
    auto decode = [&](unsigned id, int &s_, int &b_) {
        b_ = (int)(id / (unsigned)a.n_streams);
        s_ = (int)(id - (unsigned)b_ * (unsigned)a.n_streams);
    };
    unsigned *sched_slot = reinterpret_cast<unsigned *>(smem + (size_t)D * PS * sizeof(float2));
    int s = 0, bidx = 0;
    decode(cur, s, bidx);
    if (cur < total_tiles) {
        __amdgpu_buffer_rsrc_t rsrc; int voff, off;
        tile_geom(s, bidx, rsrc, voff, off);
        fetch_phases(bidx);
        fetch(rsrc, voff, -1);
    }
    STAMP_DECL;

    while (cur < total_tiles) {
        const long long n0 = (long long)bidx * NTE - OVL;     // first covered output
        int s_nxt, b_nxt;
        decode(nxt, s_nxt, b_nxt);

        STAMP(7);
        stage(s, bidx);
        STAMP(0);
        __syncthreads();
        STAMP(1);
        // next tile's HBM traffic flies under this tile's MAC loop.  (Issuing it in one
        // burst per polyphase component instead was measured slower: unrolling the phase
        // loop costs more registers than the 256-VGPR budget has.)
        __amdgpu_buffer_rsrc_t rsrc_n; int voff_n, off_n;
        tile_geom(s_nxt, b_nxt, rsrc_n, voff_n, off_n);
        if (nxt >= total_tiles) voff_n = 0x7ffff000 - NI * LANE_BYTES * TILED_THREADS;   // out of range: zeros, no traffic
        unsigned nn = nxt + G;
        if (a.sched && t == 0) nn = atomicAdd(a.sched, 1u) + 2u * G;       // arrives during the MAC loop
        STAMP(2);

        STAMP(3);
        // ---------------- MAC loop: R outputs per lane -----------------------------
        // The lane's samples of a pass (polyphase component p) come in blocks of R
        // (block b = samples bR .. bR+R-1, LDS slot offset b(R+1)).  Step k of the pass
        // multiplies taps kR .. kR+R-1 against blocks k and k+1.  Everything a step needs
        // is requested one step (R*R packed FMAs, ~256 cycles) ahead: block k+2 at the top
        // of step k and the taps of step k+1 inside it.  Three register sets take the
        // block roles in turn, so nothing is moved.
        typedef float tapvec __attribute__((ext_vector_type(R * TW)));
        static_assert(R == 8, "the MAC step below is written for 8 accumulators");
        f32x2 av[R];
#pragma unroll
        for (int r = 0; r < R; ++r) av[r] = f32x2{0.f, 0.f};
        const int lane_base = (tl + 1) * R + (tl + 1);     // slot of mm = (tl+1)R
        const int nb = Tq >> LOGR;                         // steps per pass
        tapvec hcur = *reinterpret_cast<const tapvec __attribute__((address_space(4))) *>(hp);

        // Real taps: one step is ONE asm statement -- the scalar load of the next step's
        // taps, the 64 packed FMAs (tap = one half of an SGPR pair, chosen with op_sel) and
        // the wait for that load.  The load is in flight only inside the statement, so the
        // compiler never sees registers with pending writes.  (Left to the compiler, the
        // tap load is moved next to its use and every step waits a scalar-cache latency;
        // that wait also drains the LDS reads just issued, because SMEM and LDS share a
        // counter.)  The tap table is phase-major, contiguous and padded by R taps, so the
        // load of the last step of a pass fetches the first step of the next pass.
        auto step_real = [&](const f32x2 (&cur)[R], const f32x2 (&nxt)[R], int g) {
            const f32x2 h01{hcur[0], hcur[1]}, h23{hcur[2], hcur[3]}, h45{hcur[4], hcur[5]}, h67{hcur[6], hcur[7]};
            const cfloat_p src = hp + (size_t)(g + 1) * R;
            tapvec hnext;
            asm volatile(
#include "mac_step.inc"
                : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]),
                  "+v"(av[7]), "=&s"(hnext)
                : "v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]), "v"(cur[4]), "v"(cur[5]), "v"(cur[6]),
                  "v"(cur[7]), "v"(nxt[0]), "v"(nxt[1]), "v"(nxt[2]), "v"(nxt[3]), "v"(nxt[4]), "v"(nxt[5]),
                  "v"(nxt[6]), "s"(h01), "s"(h23), "s"(h45), "s"(h67), "s"(src));
            hcur = hnext;
        };
        // Complex taps: plain C++, the compiler schedules the tap loads.
        auto step_cplx = [&](const f32x2 (&cur)[R], const f32x2 (&nxt)[R], int g) {
            const tapvec h = hcur;
#pragma unroll
            for (int i = 0; i < R * TW; ++i) hcur[i] = hp[(size_t)(g + 1) * (R * TW) + i];
#pragma unroll
            for (int qq = 0; qq < R; ++qq) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int idx = qq + r;
                    const f32x2 xv = idx < R ? cur[idx] : nxt[idx - R];
                    const float hr = h[(2 * qq) % (R * TW)], hi = h[(2 * qq + 1) % (R * TW)];
                    av[r].x = __builtin_fmaf(hr, xv.x, av[r].x);
                    av[r].x = __builtin_fmaf(-hi, xv.y, av[r].x);
                    av[r].y = __builtin_fmaf(hr, xv.y, av[r].y);
                    av[r].y = __builtin_fmaf(hi, xv.x, av[r].y);
                }
            }
        };

#pragma unroll
        for (int p = 0; p < D; ++p) {
            const f32x2 *xp = reinterpret_cast<const f32x2 *>(xs) + p * PS + lane_base;
            f32x2 wA[R], wB[R], wC[R];
            auto load_blk = [&](f32x2 (&dst)[R], int blk) {
#pragma unroll
                for (int j = 0; j < R; ++j) dst[j] = xp[blk * (R + 1) + j];
            };
            auto step = [&](const f32x2 (&cur)[R], const f32x2 (&nxt)[R], f32x2 (&ld)[R], int k) {
                load_blk(ld, k + 2);
                if (CTAPS) step_cplx(cur, nxt, p * nb + k);
                else step_real(cur, nxt, p * nb + k);
            };
            load_blk(wA, 0);
            load_blk(wB, 1);
            // this pass's share of the next tile's HBM loads, issued while the first two sample
            // blocks are on their way from LDS (p is a compile-time constant: the pass loop is unrolled)
            fetch(rsrc_n, voff_n, p);
            int k = 0;
            for (; k + 3 <= nb; k += 3) {
                step(wA, wB, wC, k);
                step(wB, wC, wA, k + 1);
                step(wC, wA, wB, k + 2);
            }
            if (k < nb) step(wA, wB, wC, k);
            if (k + 1 < nb) step(wB, wC, wA, k + 1);
        }

        float2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float2(av[r].x, av[r].y);

        STAMP(4);
        // ---------------- epilogue ---------------------------------------------------
        const long long nl = n0 + (long long)tl * R;         // first output of this lane
        if (PREMIX && !DIRECT) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = cmul_fma(acc[r], vreg[r]);
        }
        if (ROT) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = cmul_ref(acc[r], gq[r]);   // gr_rotator: z = in * d_phase
        }

        if (!DEMOD) {
            float2 *__restrict__ y = a.y_out + s * a.y_stride;
            if (a.vec_store && nl + R <= a.n_out) {
                float4 *dst = reinterpret_cast<float4 *>(y + nl);
#pragma unroll
                for (int r = 0; r < R; r += 2)
                    dst[r >> 1] = make_float4(acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (nl + r < a.n_out) y[nl + r] = acc[r];
            }
        } else {
            // previous output for r = 0: the neighbouring lane's last one.  Lane 0 of a wave
            // holds the overlap outputs: it only hands its last one on.  At the start of a
            // stream that one is the carry of the previous call (acc * v frame, tile-local
            // index R-1), or zero for a fresh block.
            if (bidx == 0 && t == 0) {
                float2 yp = a.y_prev ? a.y_prev[s] : make_float2(0.f, 0.f);
                const float2 vm = a.vtab[R - 1];
                acc[R - 1] = cmul_fma(yp, make_float2(vm.x, -vm.y));
            }
            float2 prev;
            prev.x = __shfl_up(acc[R - 1].x, 1);
            prev.y = __shfl_up(acc[R - 1].y, 1);
            float d[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                d[r] = quad_demod_fast(acc[r], prev, a.gain, s_atan);
                prev = acc[r];
            }
            float *__restrict__ o = a.d_out + s * a.d_stride;
            const bool owner = (t & 63) != 0;               // lane 0 of a wave: overlap only
            if (owner) {
                if (a.vec_store && nl + R <= a.n_out) {
                    float4 *dst = reinterpret_cast<float4 *>(o + nl);
#pragma unroll
                    for (int r = 0; r < R; r += 4) dst[r >> 2] = make_float4(d[r], d[r + 1], d[r + 2], d[r + 3]);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (nl + r < a.n_out) o[nl + r] = d[r];
                }
                // carry for the next call: the last output of the stream
                const long long last = a.n_out - 1;
                if (a.y_last && last >= nl && last < nl + R) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (nl + r == last) a.y_last[s] = cmul_fma(acc[r], a.vtab[tl * R + r]);
                }
            }
        }
        if (nxt < total_tiles) fetch_phases(b_nxt);      // for the next tile's epilogue
        if (a.sched && t == 0) sched_slot[0] = nn;
        STAMP(5);
        __syncthreads();        // xs is rewritten by the next tile
        STAMP(6);
        if (a.sched) nn = sched_slot[0];
        // (wave-uniform by construction; saying so keeps the stream's buffer descriptor in SGPRs --
        // otherwise every load is wrapped in a waterfall loop)
        nn = (unsigned)__builtin_amdgcn_readfirstlane((int)nn);
        cur = nxt; nxt = nn;
        s = s_nxt; bidx = b_nxt;
    }
    // the last workgroup out re-arms the counter for the next launch
    if (a.sched && t == 0) {
        __threadfence();
        if (atomicAdd(a.sched + 1, 1u) == G - 1) {
            a.sched[0] = 0;
            a.sched[1] = 0;
            __threadfence();
        }
    }
#ifdef GRHIP_STAMP
    if ((t & 63) == 0 && g_stamp_buf) {
        unsigned long long *o = g_stamp_buf + ((size_t)blockIdx.x * (TILED_THREADS / 64) + (t >> 6)) * 10;
        for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
        o[8] = st_begin; o[9] = stamp_now();
    }
#endif
}

static int g_num_cus = 0;

template <int D, bool CTAPS, bool PREMIX, int EPI, int FP = 0>
static int launch_tiled_inst(const FirTiledArgs &a, hipStream_t st)
{
    size_t lds = tiled_lds_bytes(D, a.Tq);
    auto kern = fir_tiled_kernel<D, CTAPS, PREMIX, EPI, FP>;
    static size_t configured = 0;   // per instantiation
    if (lds > 48 * 1024 && lds > configured) {
        GRHIP_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
        configured = lds;
    }
    if (g_num_cus == 0) {
        int dev = 0, n = 0;
        GRHIP_HIP(hipGetDevice(&dev));
        GRHIP_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        g_num_cus = n > 0 ? n : 256;
    }
    constexpr int NEW_PER_TILE = EPI == EPI_DEMOD ? TILED_NT - TILED_R * (TILED_THREADS / 64) : TILED_NT;
    const long long tiles = ((a.n_out + NEW_PER_TILE - 1) / NEW_PER_TILE) * a.n_streams;
    int wgs = tiled_wg_per_cu(D, PREMIX, EPI);
#ifdef GRHIP_DIAG       // diagnostic builds only (make variant EXTRA=-DGRHIP_DIAG): persistent workgroups per CU
    static int wg_per_cu = 0;
    if (!wg_per_cu) { const char *e = getenv("GRHIP_WGPCU"); wg_per_cu = e ? atoi(e) : -1; }
    if (wg_per_cu > 0) wgs = wg_per_cu;
#endif
    const int fit = (int)((160 * 1024) / (lds + 256));                 // what the LDS tile allows
    if (wgs > fit) wgs = fit < 1 ? 1 : fit;
    long long grid = (long long)wgs * g_num_cus;   // persistent workgroups
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TILED_THREADS), lds, st, a);
    GRHIP_HIP(hipGetLastError());
    return GRHIP_OK;
}

template <int D>
static int launch_tiled_d(bool ctaps, bool premix, int epi, const FirTiledArgs &a, hipStream_t st)
{
    if (ctaps) {
        switch (epi) {
        case EPI_NONE: return launch_tiled_inst<D, true, false, EPI_NONE>(a, st);
        case EPI_ROTATE: return launch_tiled_inst<D, true, false, EPI_ROTATE>(a, st);
        default: return fail(GRHIP_EINVAL, "complex taps have no fused demodulator epilogue");
        }
    }
    if (premix) {
        switch (epi) {
        case EPI_ROTATE: return launch_tiled_inst<D, false, true, EPI_ROTATE>(a, st);
        case EPI_DEMOD: return launch_tiled_inst<D, false, true, EPI_DEMOD>(a, st);
        default: return fail(GRHIP_EINVAL, "premix needs a rotate epilogue");
        }
    }
    if (epi != EPI_NONE) return fail(GRHIP_EINVAL, "real taps without premix have no rotator");
    if (a.fpair) {
        if constexpr (D == 2) { if (a.fpair == 1) return launch_tiled_inst<2, false, false, EPI_NONE, 1>(a, st); }
        if constexpr (D == 4) { if (a.fpair == 2) return launch_tiled_inst<4, false, false, EPI_NONE, 2>(a, st); }
        return fail(GRHIP_EINVAL, "float-pair mode needs decimation 1 or 2");
    }
    return launch_tiled_inst<D, false, false, EPI_NONE>(a, st);
}

#ifdef GRHIP_STAMP
// debug hook (not part of the ABI): where the stamp sums go; needs 10 u64 per wave
extern "C" __attribute__((visibility("default"))) int grdbg_set_stamp_buffer(void *d_buf)
{
    unsigned long long *p = (unsigned long long *)d_buf;
    GRHIP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)));
    return GRHIP_OK;
}
#endif

int launch_fir_tiled(int decim, bool ctaps, bool premix, int epi, const FirTiledArgs &a_in, int n_streams,
                     hipStream_t st)
{
    if (a_in.n_out <= 0 || n_streams <= 0) return GRHIP_OK;
    if (!tiled_supported(decim, a_in.Tq)) return fail(GRHIP_EINVAL, "tiled FIR: unsupported shape");
    // streams are addressed with 32-bit byte offsets inside a buffer descriptor
    {
        const long long item = a_in.fpair ? 4 : 8;
        if ((a_in.n_in - a_in.n_lo) * item >= (1ll << 31) - (1ll << 20))
            return fail(GRHIP_EINVAL, "tiled FIR: more than 2 GiB of input per stream in one call; call work() in pieces");
    }
    FirTiledArgs a = a_in;
    a.n_streams = n_streams;
    // the tile queue's counter round trip hides under the MAC loop of a long filter only;
    // short filters keep the static split
    if (a.Tq * decim < 128) a.sched = nullptr;
    switch (decim) {
    case 1: return launch_tiled_d<1>(ctaps, premix, epi, a, st);
    case 2: return launch_tiled_d<2>(ctaps, premix, epi, a, st);
    case 4: return launch_tiled_d<4>(ctaps, premix, epi, a, st);
    }
    return fail(GRHIP_EINVAL, "tiled FIR: unsupported decimation %d", decim);
}

}  // namespace grhip
