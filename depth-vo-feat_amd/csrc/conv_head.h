// Direct kernels for 1-4 channel 3x3 heads (conv_head.hip), dispatched from the dvf_conv2d_* entries.
#pragma once
#include "dvf_common.h"
bool dvf_head_applicable(const dvf_conv_desc *d, int nseg);          // geometry of a 1-4 channel 3x3 head (weight gradient)
bool dvf_head_fwd_applicable(const dvf_conv_desc *d, int nseg);      // ... and the direct forward kernel is the faster one
bool dvf_head_dgrad_applicable(const dvf_conv_desc *d, int nseg);    // ... and the direct dgrad kernel is the faster one
int dvf_head_fwd(const dvf_conv_desc *d, const float *in, const float *w, const float *bias, float *out, hipStream_t st);
bool dvf_head_wide_applicable(const dvf_conv_desc *d, int nseg);
int dvf_head_fwd_segs(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg, const float *w,
                      const float *bias, float *out, hipStream_t st);
int dvf_head_dgrad(const dvf_conv_desc *d, const float *dpre, const float *w, float *din, hipStream_t st, const float *mask = nullptr);
// ws (>= dvf_head_wgrad_ws_floats(d) floats): per-block partial sums + a fixed-order finish instead of float atomics
int dvf_head_wgrad(const dvf_conv_desc *d, const float *in, const float *dpre, float *dw, int accumulate, hipStream_t st,
                   float *ws = nullptr, int64_t ws_floats = 0);
int64_t dvf_head_wgrad_ws_floats(const dvf_conv_desc *d);
bool dvf_head_seg_dgrad_applicable(const dvf_conv_desc *d, int segc);
int dvf_head_seg_dgrad(const dvf_conv_desc *d, const float *dpre, const float *w, float *din, int seg_off, int segc,
                       hipStream_t st, const float *mask = nullptr);     // mask: (mask > 0) ? din : 0 (ReLU backward of the segment's producer)
// direct forward of the thin stride-2 transposed convolutions (3x3 / 4x4, 16 or 32 output channels)
bool dvf_dconvt_applicable(const dvf_conv_desc *d, int nseg);
int dvf_dconvt_fwd(const dvf_conv_desc *d, const float *in, const float *w, const float *bias, float *out, hipStream_t st);
