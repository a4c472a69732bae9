/*
 * grhip.h -- C ABI of libgrhip.so: the MI355X (gfx950) implementation of the
 * GNU Radio 3.5.0 DMR demodulation hot path.
 *
 * Plain C types only; no exceptions cross this boundary.  Every entry point
 * returns an int status (GRHIP_OK or a negative GRHIP_E*), except the *_work
 * calls which return the number of items produced (>= 0) like
 * gr_block::general_work (gnuradio-core/src/lib/runtime/gr_block.h:107-127)
 * or a negative GRHIP_E*.
 *
 * Each block type mirrors one reference block.  The citation on every group
 * names the reference interface it replaces (paths relative to the reference
 * tree).  Semantics common to all blocks:
 *
 *  - *_work(h, noutput_items, in, out): HOST pointers, same contract as
 *    gr_sync_block / gr_sync_decimator ::work: `in` points at the oldest
 *    history item, i.e. in[0 .. history-2] are items the block saw before
 *    (runtime/gr_block.h:76-84, runtime/gr_sync_block.cc:46-50), and
 *    noutput_items*decimation + history - 1 items are readable
 *    (runtime/gr_sync_decimator.cc:46-50).  The call copies to the device,
 *    runs the kernels on the handle's stream, copies back and returns when
 *    the output is in `out`.
 *  - *_work_device(..., stream): same contract with DEVICE pointers; enqueues
 *    on `stream` (a hipStream_t passed as void*; NULL = the handle's own
 *    stream) and returns without synchronising.
 *  - setters latch a new value that takes effect at the next work call, which
 *    then returns 0 items once, exactly as the reference does
 *    (filter/gr_fir_filter_XXX.cc.t:74-79,
 *     filter/gr_freq_xlating_fir_filter_XXX.cc.t:109-114).
 *  - one handle is driven by one thread at a time (thread-per-block scheduler,
 *    runtime/gr_scheduler_tpb.cc:70-77); setters may be called from another
 *    thread.
 *  - complex items are interleaved float (re, im) == gr_complex
 *    (runtime/gr_complex.h:26).
 */
#ifndef INCLUDED_GRHIP_H
#define INCLUDED_GRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define GRHIP_API __attribute__((visibility("default")))
#else
#define GRHIP_API
#endif

/* ---- status codes -------------------------------------------------------
 * The C++ block wrappers rethrow them as the exception type the reference
 * throws for the same precondition. */
#define GRHIP_OK 0
#define GRHIP_EINVAL (-1)   /* std::invalid_argument */
#define GRHIP_ERANGE (-2)   /* std::out_of_range     */
#define GRHIP_ERUNTIME (-3) /* std::runtime_error (HIP call failed) */
#define GRHIP_ENOMEM (-4)   /* std::bad_alloc        */
#define GRHIP_ENODEV (-5)   /* no usable gfx950 device / HIP runtime */
/* NOT an error: a work call's way to say WORK_DONE (runtime/gr_block.h:63-66, where it is -1).  The
 * reference's -1 would collide with GRHIP_EINVAL, so the ABI reports it as the largest int, which no
 * call can produce as an item count (grhip_head never copies more than INT_MAX - 1 items per call).
 * Every negative return is an error. */
#define GRHIP_WORK_DONE 0x7fffffff

GRHIP_API const char *grhip_strerror(int status);
/* thread-local detail of the last failing call on this thread ("" if none) */
GRHIP_API const char *grhip_last_error(void);
GRHIP_API int grhip_device_count(int *count);
/* blocks until everything enqueued on the handle-independent default work has
 * finished on `device` (hipDeviceSynchronize) */
GRHIP_API int grhip_device_synchronize(int device);
GRHIP_API const char *grhip_version(void);

/* numeric mode of the FIR-type blocks (process-wide default, may be
 * overridden per handle):
 *   GRHIP_MODE_FAST    fastest engine for the shape, own summation order (within
 *                      1e-5 relative of the reference): tiled vector-FMA kernels, the
 *                      overlap-save engine, and -- long real-tap filters at decimation
 *                      2 / 4 -- the matrix-core engine (split-binary16 MFMA, f32
 *                      accumulation; csrc/fir_mfma.hip)
 *   GRHIP_MODE_FAST_VALU  as FAST but never the matrix cores: f32 FMAs on the vector
 *                      pipes only (the north-star's "no MFMA" form; ~2x slower at 256 taps)
 *   GRHIP_MODE_FAST_REFTAPS  as FAST, and freq_xlating's matrix-core engine also reproduces the
 *                      reference's TAP-ANGLE QUANTISATION -- its composite taps are
 *                      proto[i] * exp(j * (float)(i * fwT0)), the product rounded to binary32
 *                      (filter/gr_freq_xlating_fir_filter_XXX.cc.t:79): up to 7.6e-6 rad per tap at 256
 *                      taps, which FAST's exact angles do not carry -- to first order, by a second band
 *                      matrix on the middle k-steps.  cfg2, per element of the demodulator output: 1.16e-5
 *                      against the reference's generic build and 8.9e-6 against its SSE build (the two are
 *                      9.9e-6 apart; FAST: 1.80e-5), at 0.92 of FAST's rate.  Shapes the matrix-core
 *                      engine does not take run as in FAST.
 *   GRHIP_MODE_GENERIC summation order and unfused arithmetic of
 *                      gr_fir_XXX_generic (filter/gr_fir_XXX_generic.cc.t:30-79):
 *                      bit-exact against the generic reference path
 * Error bound of the matrix-core engine (FAST, FAST_REFTAPS): every staged tile of ~2000 outputs is scaled by ONE
 * power of two taken from its largest finite sample, so |error| <= 2^-21 * sum|taps| * (largest |sample| of the
 * tile): absolute within a tile, not relative to the local signal (a quiet stretch beside a burst keeps the
 * infinity-norm tolerance, not a per-element one); a non-finite sample spoils the outputs whose window holds it and
 * at most the rest of their 16-output block. */
#define GRHIP_MODE_FAST 0
#define GRHIP_MODE_GENERIC 1
#define GRHIP_MODE_FAST_VALU 2
#define GRHIP_MODE_FAST_REFTAPS 3
GRHIP_API int grhip_set_default_mode(int mode);
GRHIP_API int grhip_get_default_mode(void);

/* ======================================================================
 * gr_fir_filter_{ccf,fff,ccc}
 *   replaces gr_make_fir_filter_XXX(int decimation, const std::vector<TAP>&)
 *   filter/gr_fir_filter_XXX.h.t:36-66, filter/gr_fir_filter_XXX.cc.t:37-88
 * kind: "ccf" | "fff" | "ccc".  taps in forward order (complex taps
 * interleaved, ntaps counts taps not floats).  history = ntaps.
 * ====================================================================== */
typedef struct grhip_fir_filter grhip_fir_filter;
GRHIP_API int grhip_fir_filter_create(grhip_fir_filter **h, const char *kind, int decimation,
                                      const float *taps, size_t ntaps, int device);
GRHIP_API void grhip_fir_filter_destroy(grhip_fir_filter *h);
GRHIP_API int grhip_fir_filter_set_taps(grhip_fir_filter *h, const float *taps, size_t ntaps);
GRHIP_API int grhip_fir_filter_set_mode(grhip_fir_filter *h, int mode);
GRHIP_API int grhip_fir_filter_history(const grhip_fir_filter *h);
GRHIP_API int grhip_fir_filter_decimation(const grhip_fir_filter *h);
GRHIP_API int grhip_fir_filter_work(grhip_fir_filter *h, int noutput_items, const void *in, void *out);
GRHIP_API int grhip_fir_filter_work_device(grhip_fir_filter *h, int noutput_items, const void *d_in,
                                           void *d_out, void *stream);
/* kernel-level seam: gr_fir_XXX::filterN / filterNdec
 * (filter/gr_fir_XXX.h.t:87-100): output[i] = filter(&input[i*decimate]);
 * ignores latched updates, never returns 0-because-updated. */
GRHIP_API int grhip_fir_filterNdec(grhip_fir_filter *h, void *output, const void *input,
                                   unsigned long n, unsigned decimate);

/* ======================================================================
 * gri_fir_filter_with_buffer_{ccf,ccc,fff}  (the FIR kernel object that owns its delay line)
 *   replaces gri_fir_filter_with_buffer_XXX(const std::vector<TAP> &taps)
 *   filter/gri_fir_filter_with_buffer_XXX.h.t:44-126, .cc.t:30-121
 * kind "ccf" | "ccc" | "fff"; taps in forward order.  filterNdec(output, input, n, decimate) takes NEW items
 * only (n * decimate of them) -- output[i] = filter(&input[i * decimate], decimate), .cc.t:110-121; filterN is
 * decimate = 1 and filter(x) is n = 1 -- and continues from the delay line the previous calls left
 * (zeros after create / set_taps, .cc.t:44-59).  GRHIP_MODE_GENERIC accumulates in the reference's order
 * (one accumulator, term after term, .cc.t:75-79): bit-exact.  A kernel-level object: set_taps acts at once.
 * ====================================================================== */
typedef struct grhip_fir_filter_with_buffer grhip_fir_filter_with_buffer;
GRHIP_API int grhip_fir_filter_with_buffer_create(grhip_fir_filter_with_buffer **h, const char *kind, const float *taps,
                                                  size_t ntaps, int device);
GRHIP_API void grhip_fir_filter_with_buffer_destroy(grhip_fir_filter_with_buffer *h);
GRHIP_API int grhip_fir_filter_with_buffer_set_taps(grhip_fir_filter_with_buffer *h, const float *taps, size_t ntaps);
GRHIP_API int grhip_fir_filter_with_buffer_set_mode(grhip_fir_filter_with_buffer *h, int mode);
GRHIP_API int grhip_fir_filter_with_buffer_ntaps(const grhip_fir_filter_with_buffer *h);
GRHIP_API int grhip_fir_filter_with_buffer_filterNdec(grhip_fir_filter_with_buffer *h, void *output, const void *input,
                                                      unsigned long n, unsigned long decimate);
GRHIP_API int grhip_fir_filter_with_buffer_filterNdec_device(grhip_fir_filter_with_buffer *h, void *d_output,
                                                             const void *d_input, unsigned long n,
                                                             unsigned long decimate, void *stream);

/* ======================================================================
 * gr_freq_xlating_fir_filter_ccc
 *   replaces gr_make_freq_xlating_fir_filter_ccc(int decimation,
 *       const std::vector<gr_complex>& taps, double center_freq, double sampling_freq)
 *   filter/gr_freq_xlating_fir_filter_XXX.h.t:64-99, .cc.t:38-123
 * history = ntaps.  Carries the gr_rotator state (filter/gr_rotator.h:29-52)
 * across calls.
 * ====================================================================== */
typedef struct grhip_freq_xlating_fir_filter_ccc grhip_freq_xlating_fir_filter_ccc;
GRHIP_API int grhip_freq_xlating_fir_filter_ccc_create(grhip_freq_xlating_fir_filter_ccc **h,
                                                       int decimation, const float *taps,
                                                       size_t ntaps, double center_freq,
                                                       double sampling_freq, int device);
GRHIP_API void grhip_freq_xlating_fir_filter_ccc_destroy(grhip_freq_xlating_fir_filter_ccc *h);
GRHIP_API int grhip_freq_xlating_fir_filter_ccc_set_center_freq(grhip_freq_xlating_fir_filter_ccc *h,
                                                               double center_freq);
GRHIP_API int grhip_freq_xlating_fir_filter_ccc_set_taps(grhip_freq_xlating_fir_filter_ccc *h,
                                                        const float *taps, size_t ntaps);
GRHIP_API int grhip_freq_xlating_fir_filter_ccc_set_mode(grhip_freq_xlating_fir_filter_ccc *h, int mode);
GRHIP_API int grhip_freq_xlating_fir_filter_ccc_history(const grhip_freq_xlating_fir_filter_ccc *h);
GRHIP_API int grhip_freq_xlating_fir_filter_ccc_work(grhip_freq_xlating_fir_filter_ccc *h,
                                                     int noutput_items, const void *in, void *out);
GRHIP_API int grhip_freq_xlating_fir_filter_ccc_work_device(grhip_freq_xlating_fir_filter_ccc *h,
                                                            int noutput_items, const void *d_in,
                                                            void *d_out, void *stream);
/* restart the stream (rotator phase 1, counter 0) without rebuilding taps:
 * what constructing a fresh block for the next capture does. */
GRHIP_API int grhip_freq_xlating_fir_filter_ccc_reset(grhip_freq_xlating_fir_filter_ccc *h);

/* ======================================================================
 * gr_quadrature_demod_cf
 *   replaces gr_make_quadrature_demod_cf(float gain)
 *   general/gr_quadrature_demod_cf.h, general/gr_quadrature_demod_cf.cc:31-62
 *   (+ gr_fast_atan2f, general/gr_fast_atan2f.cc:125-198).  history = 2.
 * ====================================================================== */
typedef struct grhip_quadrature_demod_cf grhip_quadrature_demod_cf;
GRHIP_API int grhip_quadrature_demod_cf_create(grhip_quadrature_demod_cf **h, float gain, int device);
GRHIP_API void grhip_quadrature_demod_cf_destroy(grhip_quadrature_demod_cf *h);
GRHIP_API int grhip_quadrature_demod_cf_work(grhip_quadrature_demod_cf *h, int noutput_items,
                                             const void *in, void *out);
GRHIP_API int grhip_quadrature_demod_cf_work_device(grhip_quadrature_demod_cf *h, int noutput_items,
                                                    const void *d_in, void *d_out, void *stream);

/* ======================================================================
 * Fused hier block: freq_xlating_fir_filter_ccc -> quadrature_demod_cf
 * (what tb.connect(xlating, demod) builds; one kernel, the xlating output
 * never goes to HBM).  Same arguments as the two blocks.  `in` has the
 * xlating history (ntaps-1) in front; out is float.  Carries the rotator and
 * the demodulator's previous sample across calls.
 * ====================================================================== */
typedef struct grhip_xlating_demod grhip_xlating_demod;
GRHIP_API int grhip_xlating_demod_create(grhip_xlating_demod **h, int decimation, const float *taps,
                                         size_t ntaps, double center_freq, double sampling_freq,
                                         float gain, int device);
GRHIP_API void grhip_xlating_demod_destroy(grhip_xlating_demod *h);
GRHIP_API int grhip_xlating_demod_set_mode(grhip_xlating_demod *h, int mode);
GRHIP_API int grhip_xlating_demod_reset(grhip_xlating_demod *h);
GRHIP_API int grhip_xlating_demod_history(const grhip_xlating_demod *h);
GRHIP_API int grhip_xlating_demod_work(grhip_xlating_demod *h, int noutput_items, const void *in,
                                       void *out);
GRHIP_API int grhip_xlating_demod_work_device(grhip_xlating_demod *h, int noutput_items,
                                              const void *d_in, void *d_out, void *stream);
/* n_streams independent captures in ONE launch, every one of them processed
 * like a fresh block instance (rotator phase 1, demodulator history 0): what
 * n_streams flowgraphs with identical parameters compute.  Capture s starts at
 * d_in + s*in_stride_items complex items and has NO history in front (the
 * ntaps-1 zeros a fresh flowgraph preloads are supplied by the kernel);
 * n_samples items each; output s at d_out + s*out_stride_items floats,
 * n_samples/decimation items.  Does not touch the handle's streaming state.
 * FAST modes: every shape a batched engine takes.  GRHIP_MODE_GENERIC (bit-exact against the reference's generic
 * build): decimation 1 / 2 / 4 with at least 8 taps and 2048 outputs per capture, 16-byte aligned d_in and an even
 * in_stride_items; other shapes return GRHIP_EINVAL (run them a capture at a time through work_device). */
GRHIP_API int grhip_xlating_demod_run_captures_device(grhip_xlating_demod *h, int n_streams,
                                                      size_t n_samples, const void *d_in,
                                                      size_t in_stride_items, void *d_out,
                                                      size_t out_stride_items, void *stream);

/* ======================================================================
 * digital_clock_recovery_mm_ff
 *   replaces digital_make_clock_recovery_mm_ff(float omega, float gain_omega,
 *       float mu, float gain_mu, float omega_relative_limit)
 *   gr-digital/include/digital_clock_recovery_mm_ff.h:44-92,
 *   gr-digital/lib/digital_clock_recovery_mm_ff.cc:37-139
 * general_work contract: consumes *consumed items (consume_each), returns
 * items produced; uses at most ninput_items - 8 inputs (.cc:113).
 * GRHIP_ERANGE if omega < 1 or a gain is negative (.cc:58-61).
 * ====================================================================== */
typedef struct grhip_clock_recovery_mm_ff grhip_clock_recovery_mm_ff;
GRHIP_API int grhip_clock_recovery_mm_ff_create(grhip_clock_recovery_mm_ff **h, float omega,
                                                float gain_omega, float mu, float gain_mu,
                                                float omega_relative_limit, int device);
GRHIP_API void grhip_clock_recovery_mm_ff_destroy(grhip_clock_recovery_mm_ff *h);
GRHIP_API int grhip_clock_recovery_mm_ff_forecast(const grhip_clock_recovery_mm_ff *h, int noutput_items);
GRHIP_API int grhip_clock_recovery_mm_ff_general_work(grhip_clock_recovery_mm_ff *h, int noutput_items,
                                                      int ninput_items, const float *in, float *out,
                                                      int *consumed);
/* device form: produced/consumed are written to d_counts[0], d_counts[1]
 * (device int[2]) so a following kernel can read them without a host sync */
GRHIP_API int grhip_clock_recovery_mm_ff_general_work_device(grhip_clock_recovery_mm_ff *h,
                                                             int noutput_items, int ninput_items,
                                                             const float *d_in, float *d_out,
                                                             int *d_counts, void *stream);
GRHIP_API float grhip_clock_recovery_mm_ff_mu(grhip_clock_recovery_mm_ff *h);
GRHIP_API float grhip_clock_recovery_mm_ff_omega(grhip_clock_recovery_mm_ff *h);
GRHIP_API float grhip_clock_recovery_mm_ff_gain_mu(grhip_clock_recovery_mm_ff *h);
GRHIP_API float grhip_clock_recovery_mm_ff_gain_omega(grhip_clock_recovery_mm_ff *h);
GRHIP_API int grhip_clock_recovery_mm_ff_set_gain_mu(grhip_clock_recovery_mm_ff *h, float v);
GRHIP_API int grhip_clock_recovery_mm_ff_set_gain_omega(grhip_clock_recovery_mm_ff *h, float v);
GRHIP_API int grhip_clock_recovery_mm_ff_set_mu(grhip_clock_recovery_mm_ff *h, float v);
GRHIP_API int grhip_clock_recovery_mm_ff_set_omega(grhip_clock_recovery_mm_ff *h, float v);

/* ======================================================================
 * digital_binary_slicer_fb
 *   replaces digital_make_binary_slicer_fb()
 *   gr-digital/lib/digital_binary_slicer_fb.cc:31-59
 * ====================================================================== */
typedef struct grhip_binary_slicer_fb grhip_binary_slicer_fb;
GRHIP_API int grhip_binary_slicer_fb_create(grhip_binary_slicer_fb **h, int device);
GRHIP_API void grhip_binary_slicer_fb_destroy(grhip_binary_slicer_fb *h);
GRHIP_API int grhip_binary_slicer_fb_work(grhip_binary_slicer_fb *h, int noutput_items,
                                          const float *in, unsigned char *out);
GRHIP_API int grhip_binary_slicer_fb_work_device(grhip_binary_slicer_fb *h, int noutput_items,
                                                 const float *d_in, unsigned char *d_out, void *stream);

/* ======================================================================
 * pager_slicer_fb  (the 4-level symbol decisions a 4FSK chain needs)
 *   replaces pager_make_slicer_fb(float alpha)
 *   gr-pager/lib/pager_slicer_fb.h:30-58, pager_slicer_fb.cc:34-84
 * One-pole DC tracker (d_avg = d_avg*beta + x*alpha, floats) followed by the
 * decisions {0,1,2,3} at -2, 0, +2 of the DC-free sample.  The tracker is a
 * serial float recurrence: bit-exact, one wavefront per stream.
 * ====================================================================== */
typedef struct grhip_pager_slicer_fb grhip_pager_slicer_fb;
GRHIP_API int grhip_pager_slicer_fb_create(grhip_pager_slicer_fb **h, float alpha, int device);
GRHIP_API void grhip_pager_slicer_fb_destroy(grhip_pager_slicer_fb *h);
GRHIP_API int grhip_pager_slicer_fb_work(grhip_pager_slicer_fb *h, int noutput_items,
                                         const float *in, unsigned char *out);
GRHIP_API int grhip_pager_slicer_fb_work_device(grhip_pager_slicer_fb *h, int noutput_items,
                                                const float *d_in, unsigned char *d_out, void *stream);
/* pager_slicer_fb::dc_offset() (.h:56); synchronises with the handle's last launch */
GRHIP_API int grhip_pager_slicer_fb_dc_offset(grhip_pager_slicer_fb *h, float *dc_offset);

/* ======================================================================
 * gr_unpack_k_bits_bb  (dibits -> bits ahead of the correlator)
 *   replaces gr_make_unpack_k_bits_bb(unsigned k)
 *   general/gr_unpack_k_bits_bb.cc:32-74  (gr_sync_interpolator, k outputs per input,
 *   most significant of the k bits first); GRHIP_ERANGE if k == 0 (.cc:45-46),
 *   GRHIP_EINVAL if k > 32 (the reference shifts an unsigned int by up to k-1).
 *   noutput_items must be a multiple of k (the scheduler guarantees it, .cc:72).
 * ====================================================================== */
typedef struct grhip_unpack_k_bits_bb grhip_unpack_k_bits_bb;
GRHIP_API int grhip_unpack_k_bits_bb_create(grhip_unpack_k_bits_bb **h, unsigned k, int device);
GRHIP_API void grhip_unpack_k_bits_bb_destroy(grhip_unpack_k_bits_bb *h);
GRHIP_API int grhip_unpack_k_bits_bb_work(grhip_unpack_k_bits_bb *h, int noutput_items,
                                          const unsigned char *in, unsigned char *out);
GRHIP_API int grhip_unpack_k_bits_bb_work_device(grhip_unpack_k_bits_bb *h, int noutput_items,
                                                 const unsigned char *d_in, unsigned char *d_out, void *stream);

/* ======================================================================
 * gr_stream_to_vector, gr_vector_to_streams, gr_head  (the remaining harness adapters)
 *   gr_make_stream_to_vector(size_t item_size, size_t nitems_per_block)
 *       general/gr_stream_to_vector.cc:31-60: gr_sync_decimator, work = one memcpy: grouping items into
 *       vectors moves no data;
 *   gr_make_head(size_t sizeof_stream_item, unsigned long long nitems)
 *       general/gr_head.cc:31-62: copies until nitems have passed, then work() returns WORK_DONE --
 *       GRHIP_WORK_DONE here (the reference's -1 is an error code of this ABI);
 *   gr_make_vector_to_streams(size_t item_size, size_t nstreams)
 *       general/gr_vector_to_streams.cc:31-70: item j of every input vector goes to stream j -- the data
 *       movement of gr_stream_to_streams: create it with grhip_stream_adapter_create(split = 1, ...).
 * work(): host pointers (a host memcpy, as the reference); work_device(): device pointers, the copy is
 * queued on `stream` (0 = the handle's own).  noutput_items counts OUTPUT items (vectors for
 * stream_to_vector).
 * ====================================================================== */
typedef struct grhip_copy_adapter grhip_copy_adapter;
GRHIP_API int grhip_stream_to_vector_create(grhip_copy_adapter **h, size_t item_size, size_t nitems_per_block,
                                            int device);
GRHIP_API int grhip_head_create(grhip_copy_adapter **h, size_t sizeof_stream_item, unsigned long long nitems,
                                int device);
GRHIP_API int grhip_head_reset(grhip_copy_adapter *h);
GRHIP_API void grhip_copy_adapter_destroy(grhip_copy_adapter *h);
GRHIP_API int grhip_copy_adapter_work(grhip_copy_adapter *h, int noutput_items, const void *in, void *out);
GRHIP_API int grhip_copy_adapter_work_device(grhip_copy_adapter *h, int noutput_items, const void *d_in, void *d_out,
                                             void *stream);

/* ======================================================================
 * digital_clock_recovery_mm_cc  (the complex sibling of the M&M timing loop)
 *   replaces digital_make_clock_recovery_mm_cc(float omega, float gain_omega, float mu,
 *                                              float gain_mu, float omega_relative_limit)
 *   gr-digital/include/digital_clock_recovery_mm_cc.h:44-110,
 *   gr-digital/lib/digital_clock_recovery_mm_cc.cc:37-215 (FUDGE = 16, history 3)
 * gr_block: general_work(noutput_items, ninput_items, in, out, err, &consumed) returns the items
 * produced and stores what consume_each() would get; `err` is the optional second output (the
 * clipped timing error, .cc:137-168) and may be NULL -- as in the reference its presence selects
 * the clip limit (4.0 with, 1.0 without).  GRHIP_ERANGE for omega <= 0 or negative gains (.cc:62-65).
 * Getters store into *v and return a status.
 * ====================================================================== */
typedef struct grhip_clock_recovery_mm_cc grhip_clock_recovery_mm_cc;
GRHIP_API int grhip_clock_recovery_mm_cc_create(grhip_clock_recovery_mm_cc **h, float omega, float gain_omega,
                                                float mu, float gain_mu, float omega_relative_limit, int device);
GRHIP_API void grhip_clock_recovery_mm_cc_destroy(grhip_clock_recovery_mm_cc *h);
GRHIP_API int grhip_clock_recovery_mm_cc_forecast(grhip_clock_recovery_mm_cc *h, int noutput_items);
GRHIP_API int grhip_clock_recovery_mm_cc_history(const grhip_clock_recovery_mm_cc *h);
GRHIP_API int grhip_clock_recovery_mm_cc_general_work(grhip_clock_recovery_mm_cc *h, int noutput_items,
                                                      int ninput_items, const void *in, void *out, float *err,
                                                      int *consumed);
GRHIP_API int grhip_clock_recovery_mm_cc_general_work_device(grhip_clock_recovery_mm_cc *h, int noutput_items,
                                                             int ninput_items, const void *d_in, void *d_out,
                                                             float *d_err, int *consumed, void *stream);
GRHIP_API int grhip_clock_recovery_mm_cc_mu(grhip_clock_recovery_mm_cc *h, float *v);
GRHIP_API int grhip_clock_recovery_mm_cc_omega(grhip_clock_recovery_mm_cc *h, float *v);
GRHIP_API int grhip_clock_recovery_mm_cc_gain_mu(grhip_clock_recovery_mm_cc *h, float *v);
GRHIP_API int grhip_clock_recovery_mm_cc_gain_omega(grhip_clock_recovery_mm_cc *h, float *v);
GRHIP_API int grhip_clock_recovery_mm_cc_set_mu(grhip_clock_recovery_mm_cc *h, float v);
GRHIP_API int grhip_clock_recovery_mm_cc_set_omega(grhip_clock_recovery_mm_cc *h, float v);
GRHIP_API int grhip_clock_recovery_mm_cc_set_gain_mu(grhip_clock_recovery_mm_cc *h, float v);
GRHIP_API int grhip_clock_recovery_mm_cc_set_gain_omega(grhip_clock_recovery_mm_cc *h, float v);

/* ======================================================================
 * gr_pfb_decimator_ccf  (polyphase decimator, one output channel)
 *   replaces gr_make_pfb_decimator_ccf(unsigned decim, const std::vector<float> &taps,
 *                                      unsigned channel)
 *   filter/gr_pfb_decimator_ccf.h:100-140, filter/gr_pfb_decimator_ccf.cc:43-68 (constructor),
 *   77-111 (set_taps: filter j gets taps[j + t*decim], history = taps per filter), 130-180 (work:
 *   `decim` input streams, stream s feeds filter decim-1-s, the filter outputs go through a
 *   decim-point backward FFT of which bin `channel` is the output item).
 * gr_sync_block: work() returns 0 once after set_taps (.cc:138-141).  ins[s] / the device
 * streams carry history()-1 old items in front; on the device stream s starts at
 * d_in + s * stream_stride_items complex items.
 * ====================================================================== */
typedef struct grhip_pfb_decimator_ccf grhip_pfb_decimator_ccf;
GRHIP_API int grhip_pfb_decimator_ccf_create(grhip_pfb_decimator_ccf **h, unsigned decim, const float *taps,
                                             size_t ntaps, unsigned channel, int device);
GRHIP_API void grhip_pfb_decimator_ccf_destroy(grhip_pfb_decimator_ccf *h);
GRHIP_API int grhip_pfb_decimator_ccf_set_taps(grhip_pfb_decimator_ccf *h, const float *taps, size_t ntaps);
GRHIP_API int grhip_pfb_decimator_ccf_history(const grhip_pfb_decimator_ccf *h);
GRHIP_API int grhip_pfb_decimator_ccf_work(grhip_pfb_decimator_ccf *h, int noutput_items, const void *const *ins,
                                           void *out);
GRHIP_API int grhip_pfb_decimator_ccf_work_device(grhip_pfb_decimator_ccf *h, int noutput_items, const void *d_in,
                                                  size_t stream_stride_items, void *d_out, void *stream);

/* ======================================================================
 * gr_framer_sink_1  (the consumer of the correlator's flag bit)
 *   replaces gr_make_framer_sink_1(gr_msg_queue_sptr target_queue)
 *   general/gr_framer_sink_1.h:62-98, general/gr_framer_sink_1.cc:34-66 (states), 90-190 (work):
 *   items carry the data bit in bit 0 and "first bit after the access code" in bit 1; the flagged
 *   item starts a 32-bit header (two equal 16-bit words: 4 bits whitener offset, 12 bits payload
 *   length), followed by 8 * length payload bits, most significant bit of each byte first.  Flags
 *   inside a header or payload are ignored; a header whose halves differ returns to the search.
 * The reference inserts gr_message(type 0, arg1 = whitener offset, arg2 = 0, length) into its queue
 * from inside work(); here work()/work_device() collect the messages on the device and the caller
 * moves them into its queue with message_count() + pop():
 *   message_count  waits for the queued work on `stream`, brings every complete message to the host
 *                  and returns how many are waiting to be popped;
 *   pop            next message in order: returns the payload length (0..4095), stores arg1 in
 *                  *whitener_offset and the payload in `payload` (capacity >= 4096 always fits).
 * work() returns noutput_items (a sink consumes everything, .cc:189).  State (partial header,
 * partial payload) carries across calls exactly as in the reference.
 * ====================================================================== */
typedef struct grhip_framer_sink_1 grhip_framer_sink_1;
GRHIP_API int grhip_framer_sink_1_create(grhip_framer_sink_1 **h, int device);
GRHIP_API void grhip_framer_sink_1_destroy(grhip_framer_sink_1 *h);
/* tuning: items per segment of the segment-parallel walk of long calls (0 = chosen per call, about sqrt(160 n);
 * results do not depend on it -- the tests use 64 to force the walk's divergence branch) */
GRHIP_API int grhip_framer_sink_1_set_segment_items(grhip_framer_sink_1 *h, long long items);
GRHIP_API int grhip_framer_sink_1_work(grhip_framer_sink_1 *h, int noutput_items, const unsigned char *in);
GRHIP_API int grhip_framer_sink_1_work_device(grhip_framer_sink_1 *h, int noutput_items, const unsigned char *d_in,
                                              void *stream);
GRHIP_API int grhip_framer_sink_1_message_count(grhip_framer_sink_1 *h, void *stream);
GRHIP_API int grhip_framer_sink_1_pop(grhip_framer_sink_1 *h, int *whitener_offset, unsigned char *payload,
                                      int capacity);
/* several messages at once (after message_count): offsets, lengths and the payloads back to back;
 * stops before a message whose payload would not fit; returns how many were popped */
GRHIP_API int grhip_framer_sink_1_drain(grhip_framer_sink_1 *h, int max_msgs, int *whitener_offsets, int *lengths,
                                        unsigned char *payload, size_t payload_capacity);

/* Multi-capture entry of gr_framer_sink_1 (same reference, general/gr_framer_sink_1.cc:90-190): n_streams
 * independent item streams framed by one call, one wavefront per stream, every stream from the search
 * state (a capture is framed whole, as the chain processes it; a packet cut off by the end of the
 * capture is dropped, as the reference drops it when the flowgraph ends).  Stream s is at
 * d_in + s * stream_stride_items and holds min(n_items_max, d_nitems[s * nitems_stride]) items (d_nitems,
 * a device array, may be NULL: n_items_max each) -- e.g. the d_bits / d_nbits of grhip_dmr_chain.
 *   run_device  enqueues on `stream`, no synchronisation;
 *   fetch       waits for it, brings every stream's messages to the host, returns their total number;
 *   count/get   messages of one stream, in order: get returns the payload length (0..4095), stores arg1. */
typedef struct grhip_framer_sink_1_batch grhip_framer_sink_1_batch;
GRHIP_API int grhip_framer_sink_1_batch_create(grhip_framer_sink_1_batch **h, int n_streams, size_t max_items_per_stream,
                                               int device);
GRHIP_API void grhip_framer_sink_1_batch_destroy(grhip_framer_sink_1_batch *h);
GRHIP_API int grhip_framer_sink_1_batch_run_device(grhip_framer_sink_1_batch *h, const unsigned char *d_in,
                                                   size_t stream_stride_items, const int *d_nitems, int nitems_stride,
                                                   size_t n_items_max, void *stream);
GRHIP_API int grhip_framer_sink_1_batch_fetch(grhip_framer_sink_1_batch *h, void *stream);
GRHIP_API int grhip_framer_sink_1_batch_count(grhip_framer_sink_1_batch *h, int stream_index);
GRHIP_API int grhip_framer_sink_1_batch_get(grhip_framer_sink_1_batch *h, int stream_index, int msg_index,
                                            int *whitener_offset, unsigned char *payload, int capacity);

/* ======================================================================
 * gr_stream_to_streams / gr_streams_to_stream  (the adapters either side of the
 * channeliser)
 *   replace gr_make_stream_to_streams(size_t item_size, size_t nstreams) and
 *           gr_make_streams_to_stream(size_t item_size, size_t nstreams)
 *   general/gr_stream_to_streams.cc:32-66 (gr_sync_decimator by nstreams),
 *   general/gr_streams_to_stream.cc:32-69 (gr_sync_interpolator by nstreams)
 * `split` selects the direction at creation.  work(): `streams` is the array of nstreams
 * host pointers the scheduler hands over; work_device(): stream j lives at
 * d_streams + j * stream_stride_items items (the convention of the PFB's inputs).
 * n_items_per_stream = noutput_items for stream_to_streams, noutput_items / nstreams for
 * streams_to_stream (which must divide, .cc:56).
 * ====================================================================== */
typedef struct grhip_stream_adapter grhip_stream_adapter;
GRHIP_API int grhip_stream_adapter_create(grhip_stream_adapter **h, int split, size_t item_size, size_t nstreams,
                                          int device);
GRHIP_API void grhip_stream_adapter_destroy(grhip_stream_adapter *h);
GRHIP_API int grhip_stream_adapter_work(grhip_stream_adapter *h, int n_items_per_stream, void *single,
                                        void *const *streams);
GRHIP_API int grhip_stream_adapter_work_device(grhip_stream_adapter *h, int n_items_per_stream, void *d_single,
                                               void *d_streams, size_t stream_stride_items, void *stream);

/* ======================================================================
 * digital_correlate_access_code_bb
 *   replaces digital_make_correlate_access_code_bb(const std::string&
 *       access_code, int threshold)
 *   gr-digital/include/digital_correlate_access_code_bb.h,
 *   gr-digital/lib/digital_correlate_access_code_bb.cc:37-133
 * access_code: string of '0'/'1' characters (only the LSB of each byte is
 * used, .cc:80); GRHIP_ERANGE if longer than 64 (.cc:54-57).
 * ====================================================================== */
typedef struct grhip_correlate_access_code_bb grhip_correlate_access_code_bb;
GRHIP_API int grhip_correlate_access_code_bb_create(grhip_correlate_access_code_bb **h,
                                                    const char *access_code, size_t len,
                                                    int threshold, int device);
GRHIP_API void grhip_correlate_access_code_bb_destroy(grhip_correlate_access_code_bb *h);
GRHIP_API int grhip_correlate_access_code_bb_set_access_code(grhip_correlate_access_code_bb *h,
                                                             const char *access_code, size_t len);
GRHIP_API int grhip_correlate_access_code_bb_work(grhip_correlate_access_code_bb *h, int noutput_items,
                                                  const unsigned char *in, unsigned char *out);
GRHIP_API int grhip_correlate_access_code_bb_work_device(grhip_correlate_access_code_bb *h,
                                                         int noutput_items, const unsigned char *d_in,
                                                         unsigned char *d_out, void *stream);

/* ======================================================================
 * gr_fft_vcc
 *   replaces gr_make_fft_vcc(int fft_size, bool forward,
 *       const std::vector<float>& window, bool shift)
 *   general/gr_fft_vcc.h:41-59, general/gr_fft_vcc.cc:34-64,
 *   general/gr_fft_vcc_fftw.cc:39-103 (FFTW3f c2c, unnormalised)
 * items are vectors of fft_size complex.  window: NULL/0 or fft_size floats.
 * GRHIP_ERANGE if fft_size <= 0 (general/gri_fft.cc:104-105).  Every other size the
 * reference hands to FFTW is taken (general/gri_fft.cc:97-123): powers of two up to 8192
 * by the radix-16 register kernels, larger ones (up to 2^26) in four-step form, sizes that
 * are not a power of two by a direct DFT (<= 128) or Bluestein's chirp convolution (up to
 * 2^25); beyond that GRHIP_EINVAL (a handle's work buffers are sized for 2^26 points).
 * ====================================================================== */
typedef struct grhip_fft_vcc grhip_fft_vcc;
GRHIP_API int grhip_fft_vcc_create(grhip_fft_vcc **h, int fft_size, int forward, const float *window,
                                   size_t window_len, int shift, int device);
GRHIP_API void grhip_fft_vcc_destroy(grhip_fft_vcc *h);
/* returns 1 if accepted, 0 if the length is wrong (gr_fft_vcc::set_window) */
GRHIP_API int grhip_fft_vcc_set_window(grhip_fft_vcc *h, const float *window, size_t window_len);
GRHIP_API int grhip_fft_vcc_work(grhip_fft_vcc *h, int noutput_items, const void *in, void *out);
GRHIP_API int grhip_fft_vcc_work_device(grhip_fft_vcc *h, int noutput_items, const void *d_in,
                                        void *d_out, void *stream);

/* ======================================================================
 * gr_fft_filter_ccc  
 *   replaces gr_make_fft_filter_ccc(int decimation, const std::vector<gr_complex>& taps)
 *   filter/gr_fft_filter_ccc.cc:46-128, filter/gri_fft_filter_ccc_generic.cc:63-170
 * Overlap-add fast convolution with the reference's sizes (fftsize = 2*2^ceil(log2 ntaps),
 * nsamples = fftsize - ntaps + 1 = the block's output multiple), taps pre-scaled by
 * 1/fftsize, tail carried between blocks and calls.  gr_sync_decimator, history 1.
 * noutput_items must be a multiple of nsamples (the reference asserts it, .cc:121).
 * set_taps takes effect at the next work call, which returns 0 (.cc:113-118) and clears
 * the tail (generic.cc:69-71).  Any tap count up to 2^25 (fftsize <= 2^26; beyond 4096 taps the
 * transforms run in four-step form).
 * ====================================================================== */
typedef struct grhip_fft_filter_ccc grhip_fft_filter_ccc;
GRHIP_API int grhip_fft_filter_ccc_create(grhip_fft_filter_ccc **h, int decimation, const float *taps,
                                          size_t ntaps, int device);
GRHIP_API void grhip_fft_filter_ccc_destroy(grhip_fft_filter_ccc *h);
GRHIP_API int grhip_fft_filter_ccc_set_taps(grhip_fft_filter_ccc *h, const float *taps, size_t ntaps);
GRHIP_API int grhip_fft_filter_ccc_nsamples(const grhip_fft_filter_ccc *h);   /* output multiple */
GRHIP_API int grhip_fft_filter_ccc_decimation(const grhip_fft_filter_ccc *h);
GRHIP_API int grhip_fft_filter_ccc_work(grhip_fft_filter_ccc *h, int noutput_items, const void *in, void *out);
GRHIP_API int grhip_fft_filter_ccc_work_device(grhip_fft_filter_ccc *h, int noutput_items, const void *d_in,
                                               void *d_out, void *stream);

/* ======================================================================
 * gr_pfb_channelizer_ccf
 *   replaces gr_make_pfb_channelizer_ccf(unsigned numchans,
 *       const std::vector<float>& taps, float oversample_rate)
 *   filter/gr_pfb_channelizer_ccf.h:115-178, filter/gr_pfb_channelizer_ccf.cc:36-200
 * numchans input streams, one output stream of numchans-complex vectors.
 * history = taps_per_filter + 1 on every input.  GRHIP_EINVAL if
 * numchans/oversample_rate is not an integer (.cc:57-60).
 * general_work: ins[j] points at stream j including history; returns
 * noutput_items, *consumed = items to consume on every input.
 * ====================================================================== */
typedef struct grhip_pfb_channelizer_ccf grhip_pfb_channelizer_ccf;
GRHIP_API int grhip_pfb_channelizer_ccf_create(grhip_pfb_channelizer_ccf **h, unsigned numchans,
                                               const float *taps, size_t ntaps, float oversample_rate,
                                               int device);
GRHIP_API void grhip_pfb_channelizer_ccf_destroy(grhip_pfb_channelizer_ccf *h);
GRHIP_API int grhip_pfb_channelizer_ccf_set_taps(grhip_pfb_channelizer_ccf *h, const float *taps,
                                                 size_t ntaps);
GRHIP_API int grhip_pfb_channelizer_ccf_history(const grhip_pfb_channelizer_ccf *h);
GRHIP_API int grhip_pfb_channelizer_ccf_output_multiple(const grhip_pfb_channelizer_ccf *h);
GRHIP_API int grhip_pfb_channelizer_ccf_general_work(grhip_pfb_channelizer_ccf *h, int noutput_items,
                                                     const void *const *ins, void *out, int *consumed);
/* device form: the numchans streams live in ONE device buffer, stream j at
 * d_in + j*stream_stride_items complex items */
GRHIP_API int grhip_pfb_channelizer_ccf_general_work_device(grhip_pfb_channelizer_ccf *h,
                                                            int noutput_items, const void *d_in,
                                                            size_t stream_stride_items, void *d_out,
                                                            void *stream);

/* The hier block blks2.pfb_channelizer_ccf (gnuradio-core/src/python/gnuradio/blks2impl/pfb_channelizer.py:25-75:
 * gr_stream_to_streams -> gr_pfb_channelizer_ccf -> gr_vector_to_streams) as ONE call on the handle above: ONE
 * interleaved input stream (general/gr_stream_to_streams.cc:57-63: stream j's item m is d_in[m * numchans + j]), numchans
 * output streams (general/gr_vector_to_streams.cc:57-63: channel k's stream at d_out + k * out_stride_items, item t is
 * bin k of output vector t).  d_in carries taps_per_filter * numchans history items in front (the channeliser's history
 * of taps_per_filter + 1 on each of its inputs), zeros at the start of a flowgraph.  noutput_items and the return value are
 * the inner block's (output vectors = items per output stream); results equal the three blocks run one after the other
 * bit for bit.  Oversample rate 1 with 2 / 4 / 8 / 16 channels takes one fused kernel (16 B of memory traffic per
 * sample instead of 48); every other shape runs the three kernels on work buffers of the handle. */
GRHIP_API int grhip_pfb_channelizer_ccf_hier_work_device(grhip_pfb_channelizer_ccf *h, int noutput_items,
                                                         const void *d_in, void *d_out, size_t out_stride_items,
                                                         void *stream);

/* ======================================================================
 * Full DMR chain as one device-resident pipeline (hier block):
 *   freq_xlating_fir_filter_ccc -> quadrature_demod_cf ->
 *   clock_recovery_mm_ff -> binary_slicer_fb -> correlate_access_code_bb
 * for n_streams independent captures with identical parameters (SURVEY 8(e):
 * streams are independent units; this is the multi-stream batch that fills
 * the device for the serial M&M stage).  Each stream starts from fresh block
 * state on every run() (a capture is processed whole).
 * Decimation 1 / 2 / 4 with any taps; every other decimation up to 256 with a
 * real prototype (imaginary parts zero) of up to 1024 taps -- create fails with
 * GRHIP_EINVAL for a shape no batched engine takes.
 * ====================================================================== */
typedef struct grhip_dmr_chain grhip_dmr_chain;
typedef struct grhip_dmr_chain_params {
    int decimation;
    const float *taps; /* complex prototype taps, interleaved */
    size_t ntaps;
    double center_freq, sampling_freq;
    float demod_gain;
    float omega, gain_omega, mu, gain_mu, omega_relative_limit;
    const char *access_code;
    size_t access_code_len;
    int threshold;
} grhip_dmr_chain_params;
GRHIP_API int grhip_dmr_chain_create(grhip_dmr_chain **h, const grhip_dmr_chain_params *p,
                                     int n_streams, size_t max_samples_per_stream, int device);
GRHIP_API void grhip_dmr_chain_destroy(grhip_dmr_chain *h);
/* GRHIP_MODE_GENERIC runs the xlating FIR in gr_fir_ccc_generic order (one
 * stream at a time): the whole chain is then bit-exact against the reference's
 * generic path, symbols and bit decisions included. */
GRHIP_API int grhip_dmr_chain_set_mode(grhip_dmr_chain *h, int mode);
/* Scheduling of the clock recovery (digital_clock_recovery_mm_ff.cc:116-134, a serial recurrence per capture): one
 * wavefront per capture (1), eight captures per wavefront (8: the shape for batches of more than a thousand
 * captures, where the loop then leaves the FIR its full grid), or thirty-two (32: two lanes per capture, the samples
 * through a FIFO in registers; the loop on a sixteenth of the CUs -- measured slower per symbol, DESIGN 4.3, an option);
 * 0 = chosen by n_streams (the default: 1 or 8).  Results are identical (every form is bit-exact on its input). */
GRHIP_API int grhip_dmr_chain_set_captures_per_wave(grhip_dmr_chain *h, int captures);
/* Upper bound on the symbols the clock recovery produces per capture: the noutput_items of its general_work
 * (digital_clock_recovery_mm_ff.cc:113, `oo < noutput_items`); 0 = no bound but the output rows (the default).
 * A capture stops at exactly that many symbols, whichever form of the loop runs. */
GRHIP_API int grhip_dmr_chain_set_max_symbols(grhip_dmr_chain *h, size_t max_symbols);
/* 4FSK tail: with enable != 0 the symbols go through pager_slicer_fb(alpha)
 * (gr-pager/lib/pager_slicer_fb.cc:47-84) -> gr_unpack_k_bits_bb(2) (general/gr_unpack_k_bits_bb.cc:64-69) ->
 * the access-code correlator instead of the binary slicer: both bits of every symbol, most significant first.
 * d_bits then receives TWO items per symbol (bits_stride >= 2 * n_samples / decimation) and d_nbits their
 * number per stream.  The access code is matched against that dibit stream (gr-digital/python/pkt.py:143-147
 * feeds the correlator unpacked bits in the same way). */
GRHIP_API int grhip_dmr_chain_set_four_level(grhip_dmr_chain *h, int enable, float pager_alpha);
/* d_in: n_streams captures of n_samples complex each, stream s at
 * d_in + s*stream_stride_items (NO history in front: the chain supplies the
 * zeros a fresh flowgraph would).  d_bits: n_streams * bits_stride bytes;
 * d_nbits: n_streams ints (symbols produced per stream). */
GRHIP_API int grhip_dmr_chain_run_device(grhip_dmr_chain *h, const void *d_in, size_t n_samples,
                                         size_t stream_stride_items, unsigned char *d_bits,
                                         size_t bits_stride, int *d_nbits, void *stream);
/* intermediate products of the last run (device pointers owned by the handle):
 * which: 0 demod floats, 1 M&M soft symbols, 2 pager_slicer symbols (bytes, 4FSK tail only);
 * *stride receives the per-stream stride in items */
GRHIP_API int grhip_dmr_chain_intermediate(grhip_dmr_chain *h, int which, void **d_ptr, size_t *stride);

/* ---- small device-memory helpers for hosts without a HIP binding -------- */
GRHIP_API int grhip_malloc(void **d_ptr, size_t bytes, int device);
GRHIP_API int grhip_free(void *d_ptr);
GRHIP_API int grhip_memcpy_h2d(void *d_dst, const void *src, size_t bytes);
GRHIP_API int grhip_memcpy_d2h(void *dst, const void *d_src, size_t bytes);
GRHIP_API int grhip_stream_synchronize(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* INCLUDED_GRHIP_H */
