// digital_kernels.h -- launchers for clock_recovery_mm_ff, binary_slicer_fb,
// correlate_access_code_bb (internal).
#pragma once
#include <hip/hip_runtime.h>

namespace grhip {

// device-resident state of one digital_clock_recovery_mm_ff instance
// (gr-digital/include/digital_clock_recovery_mm_ff.h:77-92)
struct MMState {
    float mu, omega, min_omega, omega_mid, max_omega;
    float gain_omega, gain_mu, last_sample, omega_relative_limit;
    int pad;
};

// one wave per stream; stream s reads in + s*in_stride, writes out + s*out_stride,
// state[s], counts[2*s] = produced, counts[2*s+1] = consumed.
// resume != 0: counts[] hold the stream's totals so far, the call continues from there with
// noutput_items / ninput_items counted from the stream's start (see mm_kernel).
// rows != 0: eight streams per wave (mm_rows_kernel; 16-byte aligned rows with 3 floats of slack behind ninput_items);
// rows >= 1024: with the ring of 1024 samples (36 KB of LDS per wave), else 512 (20 KB); rows == 32: thirty-two streams per
// wave (mm_pairs_kernel).  counts_out: where the totals go instead of counts[] (counts[] is then only read).
int launch_mm(MMState *state, int n_streams, int noutput_items, int ninput_items, const float *in,
              long long in_stride, float *out, long long out_stride, int *counts,
              const float *mmse_rev, hipStream_t st, int resume = 0, int rows = 0, int *counts_out = nullptr);

int launch_binary_slicer(const float *in, unsigned char *out, long long n, hipStream_t st);
// pager_slicer_fb: d_avg[s] carried in device memory; streams s at in + s*in_stride / out + s*out_stride
int launch_pager_slicer(float *d_avg, int n_streams, float alpha, float beta, const float *in, long long in_stride,
                        unsigned char *out, long long out_stride, long long n, hipStream_t st, const int *n_ptr = nullptr,
                        int n_ptr_stride = 0, int *pos = nullptr);       // pos[s]: items sliced so far, the call resumes there
int launch_unpack_k_bits_streams(unsigned k, int n_streams, const unsigned char *in, long long in_stride, unsigned char *out,
                                 long long out_stride, long long n_in_max, const int *n_ptr, int n_ptr_stride, int *n_out,
                                 int n_out_stride, hipStream_t st);
// gr_stream_to_streams (split = true) / gr_streams_to_stream (split = false): stream j of `multi` starts at
// multi + j * multi_stride_items items
int launch_streams(bool split, void *single, void *multi, long long multi_stride_items, int nstreams, size_t item_size,
                   long long n_items_per_stream, hipStream_t st);
int launch_unpack_k_bits(unsigned k, const unsigned char *in, unsigned char *out, long long noutput_items, hipStream_t st);

// device-resident state of one digital_correlate_access_code_bb instance
struct CorrState {
    unsigned long long data_reg, flag_reg;
};
struct CorrParams {
    unsigned long long access_code, mask;
    unsigned threshold;
    unsigned len;
};
// in_bytes != nullptr: byte input (LSB used); else in_soft: float input sliced
// with x >= 0 (binary_slicer fused in front).  n_ptr (optional): per-stream item
// count on the device (n = min(n, n_ptr[s*n_ptr_stride])).
int launch_correlate(const CorrParams &p, CorrState *state, int n_streams, const unsigned char *in_bytes,
                     const float *in_soft, long long in_stride, unsigned char *out, long long out_stride,
                     long long n, const int *n_ptr, int n_ptr_stride, hipStream_t st, long long n_expect = 0);

}  // namespace grhip
