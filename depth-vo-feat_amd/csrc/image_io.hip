// GPU side of the input pipeline: the reference decodes a frame, converts it to float32 and calls
// scipy.misc.imresize(img, (H, W)) (un_dataset.py:63-66, dataset.py:50-51), i.e. bytescale ([min, max] -> [0, 255], uint8)
// followed by PIL's BILINEAR resize of the uint8 image.  These kernels do the same on the device from the raw uint8
// frame, bit for bit: PIL's separable resampling with 22-bit fixed-point coefficients (Pillow src/libImaging/Resample.c,
// ImagingResampleHorizontal_8bpc / Vertical_8bpc: accumulate int32 from 1 << 21, shift by 22, clip to 0..255), horizontal
// pass first into a uint8 intermediate.  The coefficient tables are computed on the host exactly as precompute_coeffs /
// normalize_coeffs_8bpc do (depth-vo-feat_amd/dvf/image_ops.py) and cached per (input size, output size).
#include "dvf_common.h"

namespace {

__global__ __launch_bounds__(256) void minmax_u8_kernel(const uint8_t *src, int64_t n, int *mm) {
    int lo = 255, hi = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int v = src[i];
        lo = min(lo, v);
        hi = max(hi, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, 64));
        hi = max(hi, __shfl_xor(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

// scipy.misc.bytescale of the float32 image with cmin = min, cmax = max (fp32 arithmetic as numpy does it)
__device__ __forceinline__ int bytescale1(int v, float cmin, float s) {
    float f = __fmul_rn(__fsub_rn((float)v, cmin), s);
    f = fminf(fmaxf(f, 0.f), 255.f) + 0.5f;
    return (int)f;
}

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[y][ox][c] = clip8(2^21 + sum_x bytescale(src[y][xmin+x][c]) * k[ox][x])
__global__ __launch_bounds__(256) void hresample_kernel(const uint8_t *src, uint8_t *tmp, const int *bounds, const int *kk, int ksize,
                                                        int IH, int IW, int OW, int C, const int *mm) {
    const int64_t total = (int64_t)IH * OW * C;
    const float cmin = (float)mm[0];
    const float cscale = (mm[1] - mm[0]) == 0 ? 1.f : (float)(mm[1] - mm[0]);
    const float s = (float)(255.0 / (double)cscale);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C), ox = (int)((i / C) % OW), y = (int)(i / ((int64_t)C * OW));
        const int xmin = bounds[2 * ox], xn = bounds[2 * ox + 1];
        const int *k = kk + (int64_t)ox * ksize;
        const uint8_t *row = src + ((int64_t)y * IW + xmin) * C + c;
        int acc = 1 << 21;
        for (int x = 0; x < xn; ++x) acc += bytescale1(row[(int64_t)x * C], cmin, s) * k[x];
        tmp[i] = (uint8_t)clip8(acc);
    }
}

// vertical pass, written as float32 CHW: dst[c][oy][ox] = clip8(2^21 + sum_y tmp[ymin+y][ox][c] * k[oy][y])
__global__ __launch_bounds__(256) void vresample_kernel(const uint8_t *tmp, float *dst, const int *bounds, const int *kk, int ksize,
                                                        int OH, int OW, int C) {
    const int64_t total = (int64_t)C * OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), c = (int)(i / ((int64_t)OW * OH));
        const int ymin = bounds[2 * oy], yn = bounds[2 * oy + 1];
        const int *k = kk + (int64_t)oy * ksize;
        const uint8_t *col = tmp + ((int64_t)ymin * OW + ox) * C + c;
        int acc = 1 << 21;
        for (int y = 0; y < yn; ++y) acc += (int)col[(int64_t)y * OW * C] * k[y];
        dst[i] = (float)clip8(acc);
    }
}

}  // namespace

extern "C" {

int dvf_imresize_u8(const uint8_t *src_hwc, int IH, int IW, int C, const int *hbounds, const int *hcoef, int hksize,
                    const int *vbounds, const int *vcoef, int vksize, uint8_t *tmp, int *minmax, float *dst_chw, int OH, int OW,
                    void *stream) {
    if (!src_hwc || !hbounds || !hcoef || !vbounds || !vcoef || !tmp || !minmax || !dst_chw || IH <= 0 || IW <= 0 || C <= 0 ||
        OH <= 0 || OW <= 0 || hksize <= 0 || vksize <= 0)
        return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    const int init[2] = {255, 0};
    if (hipMemcpyAsync(minmax, init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) return DVF_ERR_LAUNCH;
    const int64_t n = (int64_t)IH * IW * C;
    minmax_u8_kernel<<<(int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), 256, 0, st>>>(src_hwc, n, minmax);
    DVF_LAUNCH_CHECK();
    const int64_t nh = (int64_t)IH * OW * C, nv = (int64_t)C * OH * OW;
    hresample_kernel<<<(int)((nh + 255) / 256 > 4096 ? 4096 : (nh + 255) / 256), 256, 0, st>>>(src_hwc, tmp, hbounds, hcoef, hksize, IH, IW,
                                                                                             OW, C, minmax);
    DVF_LAUNCH_CHECK();
    vresample_kernel<<<(int)((nv + 255) / 256 > 4096 ? 4096 : (nv + 255) / 256), 256, 0, st>>>(tmp, dst_chw, vbounds, vcoef, vksize, OH, OW, C);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

}  // extern "C"
