// Memory-bound helpers of the training step: bilinear x2 upsampling (DispNetS disparity pyramid),
// reciprocal (disparity -> depth), spatial mean (pose head), area down-sampling (sfm image pyramid),
// bilinear x0.5 (FeatExtractor image pyramid) and the fused multi-tensor Adam update.
#include "dvf_common.h"

namespace {

// aten area_pixel_compute_source_index(scale, dst, align_corners=False, cubic=False)
__device__ __forceinline__ void up_src(int dst, float scale, int in_size, int &i0, int &i1, float &lam) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;                       // floor: src >= 0
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    lam = src - (float)i0;
    if (lam < 0.f) lam = 0.f;
    if (lam > 1.f) lam = 1.f;
}

// out[n][c][Y][X] for Y < OH, X < OW (OH <= 2H: the crop of crop_like is folded in).  DispNetS.py:115
__global__ void resize_bilinear_fwd_kernel(const float *in, float *out, int H, int W, int OH, int OW, float sy, float sx,
                                           int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % OW), Y = (int)((i / OW) % OH);
        const int64_t plane = i / ((int64_t)OW * OH);
        int y0, y1, x0, x1;
        float ly, lx;
        up_src(Y, sy, H, y0, y1, ly);
        up_src(X, sx, W, x0, x1, lx);
        const float *p = in + plane * H * W;
        const float top = (1.f - lx) * p[(int64_t)y0 * W + x0] + lx * p[(int64_t)y0 * W + x1];
        const float bot = (1.f - lx) * p[(int64_t)y1 * W + x0] + lx * p[(int64_t)y1 * W + x1];
        out[i] = (1.f - ly) * top + ly * bot;
    }
}

// Gather-form backward for the x2 case: input pixel (y, x) collects from output rows 2y-1 .. 2y+2.
__global__ void upsample2x_bwd_kernel(const float *gout, float *gin, int H, int W, int OH, int OW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int x, y;
        int64_t plane;
        if (total < ((int64_t)1 << 31)) {      // 32-bit index arithmetic (three 64-bit divisions per element were most of the kernel)
            const unsigned i32 = (unsigned)i, row = i32 / (unsigned)W;
            x = (int)(i32 - row * (unsigned)W);
            plane = row / (unsigned)H;
            y = (int)(row - (unsigned)plane * (unsigned)H);
        } else {
            x = (int)(i % W); y = (int)((i / W) % H);
            plane = i / ((int64_t)W * H);
        }
        const float *g = gout + plane * OH * OW;
        float wy[4], wx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int Y = 2 * y - 1 + k, X = 2 * x - 1 + k;
            wy[k] = 0.f;
            wx[k] = 0.f;
            int a0, a1;
            float l;
            if (Y >= 0 && Y < OH) {
                up_src(Y, 0.5f, H, a0, a1, l);
                wy[k] = (a0 == y ? 1.f - l : 0.f) + (a1 == y ? l : 0.f);
            }
            if (X >= 0 && X < OW) {
                up_src(X, 0.5f, W, a0, a1, l);
                wx[k] = (a0 == x ? 1.f - l : 0.f) + (a1 == x ? l : 0.f);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            if (wy[ky] == 0.f) continue;
            const int Y = 2 * y - 1 + ky;
            float r = 0.f;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
                if (wx[kx] != 0.f) r += wx[kx] * g[(int64_t)Y * OW + (2 * x - 1 + kx)];
            s += wy[ky] * r;
        }
        gin[i] = s;
    }
}

__global__ void recip_fwd_kernel(const float *x, float *y, float eps, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = 1.f / (x[i] + eps);
}

__global__ void recip_bwd_kernel(const float *gy, const float *y, float *gx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        gx[i] = -gy[i] * y[i] * y[i];
}

// out[plane] = scale * mean(in[plane][:])   (PoseExpNet: pose.mean(3).mean(2) * 0.01)
__global__ __launch_bounds__(64) void spatial_mean_kernel(const float *in, float *out, int HW, float scale) {
    const float *p = in + (int64_t)blockIdx.x * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 64) s += p[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] = s * scale / (float)HW;
}

__global__ void spatial_mean_bwd_kernel(const float *gout, float *gin, int HW, float scale, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        gin[i] = gout[i / HW] * scale / (float)HW;
}

// adaptive average pooling = F.interpolate(mode='area')  (loss_functions_sfm.py:18-19)
__global__ void area_down_kernel(const float *in, float *out, int H, int W, int OH, int OW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % OW), Y = (int)((i / OW) % OH);
        const int64_t plane = i / ((int64_t)OW * OH);
        const int y0 = (int)(((int64_t)Y * H) / OH), y1 = (int)((((int64_t)Y + 1) * H + OH - 1) / OH);
        const int x0 = (int)(((int64_t)X * W) / OW), x1 = (int)((((int64_t)X + 1) * W + OW - 1) / OW);
        const float *p = in + plane * H * W;
        float s = 0.f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) s += p[(int64_t)y * W + x];
        out[i] = s / (float)((y1 - y0) * (x1 - x0));
    }
}

// torch.optim.Adam (amsgrad off, L2-in-gradient weight decay) on one flat arena.  The step counter and the
// learning rate live in DEVICE memory (opt_state = {step, lr, step_size, bc2_sqrt}) so that a captured HIP graph
// replays correctly: nothing step-dependent is baked into kernel arguments.
__global__ void adam_prep_kernel(float *opt_state, float beta1, float beta2) {
    const float t = opt_state[0] + 1.f;
    opt_state[0] = t;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    opt_state[2] = (float)((double)opt_state[1] / bc1);     // step_size = lr / (1 - beta1^t)
    opt_state[3] = (float)sqrt(bc2);                        // sqrt(1 - beta2^t)
}

__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, int64_t n, float beta1, float beta2, float eps, float wd,
                            const float *__restrict__ opt_state, float grad_scale) {
    const float step_size = opt_state[2], bc2_sqrt = opt_state[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * grad_scale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

// explainability regulariser: BCE(mask, 1) = -mean(max(log m, -100))   (loss_functions_sfm.py:49-56)
__global__ __launch_bounds__(256) void bce_ones_fwd_kernel(const float *m, float *partials, int64_t n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s -= fmaxf(logf(m[i]), -100.f);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float *partials, int nblk, float scale, float *out,
                                                           int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
        out[0] = accumulate ? out[0] + t : t;
    }
}

__global__ void bce_ones_bwd_kernel(const float *m, const float *grad_loss, float *gm, int64_t n) {
    const float g = grad_loss[0] / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        gm[i] = (logf(m[i]) > -100.f) ? -g / m[i] : 0.f;
}

// Depthwise ConvTranspose2d(C, C, k=4, stride=2, pad=1, groups=C) (+ residual add): feat_extractor.py:38-41,72-82.
// out[n][c][Y][X] = bias[c] + skip[..] + sum_{a,b : (Y+1-a), (X+1-b) even} x[n][c][(Y+1-a)/2][(X+1-b)/2] * w[c][a][b]
__global__ void dwconvt_fwd_kernel(const float *x, const float *w, const float *bias, const float *skip, float *out,
                                   int C, int H, int W, int64_t total) {
    const int OH = 2 * H, OW = 2 * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % OW), Y = (int)((i / OW) % OH);
        const int64_t plane = i / ((int64_t)OW * OH);
        const int c = (int)(plane % C);
        const float *xp = x + plane * H * W;
        const float *wp = w + c * 16;
        float s = bias ? bias[c] : 0.f;
        // a has the parity of Y+1: a in {a0, a0+2}
        const int a0 = (Y + 1) & 1, b0 = (X + 1) & 1;
#pragma unroll
        for (int da = 0; da < 2; ++da) {
            const int a = a0 + 2 * da, iy = (Y + 1 - a) >> 1;
            if (iy < 0 || iy >= H || (Y + 1 - a) < 0) continue;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int b = b0 + 2 * db, ix = (X + 1 - b) >> 1;
                if (ix < 0 || ix >= W || (X + 1 - b) < 0) continue;
                s += xp[(int64_t)iy * W + ix] * wp[a * 4 + b];
            }
        }
        if (skip) s += skip[i];
        out[i] = s;
    }
}

// dx[n][c][i][j] = sum_{a,b} dy[n][c][2i-1+a][2j-1+b] * w[c][a][b]
__global__ void dwconvt_dgrad_kernel(const float *dy, const float *w, float *dx, int C, int H, int W, int64_t total) {
    const int OH = 2 * H, OW = 2 * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xj = (int)(i % W), yi = (int)((i / W) % H);
        const int64_t plane = i / ((int64_t)W * H);
        const int c = (int)(plane % C);
        const float *gp = dy + plane * OH * OW;
        const float *wp = w + c * 16;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int Y = 2 * yi - 1 + a;
            if (Y < 0 || Y >= OH) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int X = 2 * xj - 1 + b;
                if (X >= 0 && X < OW) s += gp[(int64_t)Y * OW + X] * wp[a * 4 + b];
            }
        }
        dx[i] = s;
    }
}

// dw[c][a][b] += sum_{i,j} x[n][c][i][j] * dy[n][c][2i-1+a][2j-1+b];  db[c] += sum dy.  One block per (plane, chunk).
__global__ __launch_bounds__(256) void dwconvt_wgrad_kernel(const float *x, const float *dy, float *dw, float *db, int C,
                                                            int H, int W, int chunks) {
    __shared__ float red[4][17];
    const int OH = 2 * H, OW = 2 * W;
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const int c = plane % C;
    const float *xp = x + (int64_t)plane * H * W;
    const float *gp = dy + (int64_t)plane * OH * OW;
    const int per = (H * W + chunks - 1) / chunks, beg = chunk * per, end = min(H * W, beg + per);
    float acc[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) acc[k] = 0.f;
    for (int p = beg + threadIdx.x; p < end; p += 256) {
        const int yi = p / W, xj = p - yi * W;
        const float xv = xp[p];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int Y = 2 * yi - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int X = 2 * xj - 1 + b;
                if (Y >= 0 && Y < OH && X >= 0 && X < OW) acc[a * 4 + b] += xv * gp[(int64_t)Y * OW + X];
            }
        }
        // every output pixel belongs to exactly one (i, j) through its (even, even)-offset 2x2 cell: Y in {2i, 2i+1}
        acc[16] += gp[(int64_t)(2 * yi) * OW + 2 * xj] + gp[(int64_t)(2 * yi) * OW + 2 * xj + 1] +
                   gp[(int64_t)(2 * yi + 1) * OW + 2 * xj] + gp[(int64_t)(2 * yi + 1) * OW + 2 * xj + 1];
    }
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        const float r = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 17) {
        const float r = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (threadIdx.x < 16) atomicAdd(&dw[c * 16 + threadIdx.x], r);
        else if (db) atomicAdd(&db[c], r);
    }
}

inline int nblocks(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int dvf_resize_bilinear_fwd(const float *in, float *out, int planes, int H, int W, int OH, int OW, float scale_h,
                            float scale_w, void *stream) {
    if (!in || !out || planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return DVF_ERR_INVALID_ARG;
    const int64_t total = (int64_t)planes * OH * OW;
    resize_bilinear_fwd_kernel<<<nblocks(total), 256, 0, dvf_stream(stream)>>>(in, out, H, W, OH, OW, scale_h, scale_w,
                                                                              total);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_upsample2x_bwd(const float *gout, float *gin, int planes, int H, int W, int OH, int OW, void *stream) {
    if (!gout || !gin || planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || OH > 2 * H || OW > 2 * W)
        return DVF_ERR_INVALID_ARG;
    const int64_t total = (int64_t)planes * H * W;
    upsample2x_bwd_kernel<<<nblocks(total), 256, 0, dvf_stream(stream)>>>(gout, gin, H, W, OH, OW, total);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_recip_fwd(const float *x, float *y, float eps, int64_t n, void *stream) {
    if (!x || !y || n <= 0) return DVF_ERR_INVALID_ARG;
    recip_fwd_kernel<<<nblocks(n), 256, 0, dvf_stream(stream)>>>(x, y, eps, n);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_recip_bwd(const float *gy, const float *y, float *gx, int64_t n, void *stream) {
    if (!gy || !y || !gx || n <= 0) return DVF_ERR_INVALID_ARG;
    recip_bwd_kernel<<<nblocks(n), 256, 0, dvf_stream(stream)>>>(gy, y, gx, n);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_spatial_mean_fwd(const float *in, float *out, int planes, int HW, float scale, void *stream) {
    if (!in || !out || planes <= 0 || HW <= 0) return DVF_ERR_INVALID_ARG;
    spatial_mean_kernel<<<planes, 64, 0, dvf_stream(stream)>>>(in, out, HW, scale);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_spatial_mean_bwd(const float *gout, float *gin, int planes, int HW, float scale, void *stream) {
    if (!gout || !gin || planes <= 0 || HW <= 0) return DVF_ERR_INVALID_ARG;
    const int64_t total = (int64_t)planes * HW;
    spatial_mean_bwd_kernel<<<nblocks(total), 256, 0, dvf_stream(stream)>>>(gout, gin, HW, scale, total);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_area_downsample(const float *in, float *out, int planes, int H, int W, int OH, int OW, void *stream) {
    if (!in || !out || planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return DVF_ERR_INVALID_ARG;
    const int64_t total = (int64_t)planes * OH * OW;
    area_down_kernel<<<nblocks(total), 256, 0, dvf_stream(stream)>>>(in, out, H, W, OH, OW, total);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_dwconvt4x4s2_fwd(const float *x, const float *w, const float *bias, const float *skip, float *out, int N, int C,
                         int H, int W, void *stream) {
    if (!x || !w || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DVF_ERR_INVALID_ARG;
    const int64_t total = (int64_t)N * C * 4 * H * W;
    dwconvt_fwd_kernel<<<nblocks(total), 256, 0, dvf_stream(stream)>>>(x, w, bias, skip, out, C, H, W, total);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_dwconvt4x4s2_bwd(const float *x, const float *w, const float *dy, float *dx, float *dw, float *db, int N, int C,
                         int H, int W, void *stream) {
    if (!x || !w || !dy || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    if (dx) {
        const int64_t total = (int64_t)N * C * H * W;
        dwconvt_dgrad_kernel<<<nblocks(total), 256, 0, st>>>(dy, w, dx, C, H, W, total);
        DVF_LAUNCH_CHECK();
    }
    if (dw) {
        if (hipMemsetAsync(dw, 0, sizeof(float) * 16 * C, st) != hipSuccess) return DVF_ERR_LAUNCH;
        if (db && hipMemsetAsync(db, 0, sizeof(float) * C, st) != hipSuccess) return DVF_ERR_LAUNCH;
        int chunks = (H * W + 2047) / 2048;
        while (chunks > 1 && (int64_t)N * C * chunks > 8192) chunks >>= 1;
        dwconvt_wgrad_kernel<<<N * C * chunks, 256, 0, st>>>(x, dy, dw, db, C, H, W, chunks);
        DVF_LAUNCH_CHECK();
    }
    return DVF_OK;
}

int dvf_bce_ones_fwd(const float *mask, float *loss_out, float *partials, int64_t n, int accumulate, void *stream) {
    if (!mask || !loss_out || !partials || n <= 0) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    const int nb = nblocks(n) > 1024 ? 1024 : nblocks(n);
    bce_ones_fwd_kernel<<<nb, 256, 0, st>>>(mask, partials, n);
    DVF_LAUNCH_CHECK();
    sum_partials_kernel<<<1, 256, 0, st>>>(partials, nb, 1.f / (float)n, loss_out, accumulate);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_bce_ones_bwd(const float *mask, const float *grad_loss, float *g_mask, int64_t n, void *stream) {
    if (!mask || !grad_loss || !g_mask || n <= 0) return DVF_ERR_INVALID_ARG;
    bce_ones_bwd_kernel<<<nblocks(n), 256, 0, dvf_stream(stream)>>>(mask, grad_loss, g_mask, n);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float *opt_state,
                  int advance_step, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !opt_state || n <= 0) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    if (advance_step) {
        adam_prep_kernel<<<1, 1, 0, st>>>(opt_state, beta1, beta2);
        DVF_LAUNCH_CHECK();
    }
    adam_kernel<<<nblocks(n), 256, 0, st>>>(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, weight_decay,
                                            opt_state, grad_scale);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

}  // extern "C"
