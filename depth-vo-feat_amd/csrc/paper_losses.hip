// Paper-faithful loss variant from the reference's Caffe training graph: edge-aware first-order
// smoothness of the inverse depth, experiments/depth_odometry_feature/train.prototxt:4452-4661.
//
//   gI_v = |EdgeX * I_c| (3x3 valid cross-correlation, filler.hpp:267-288: 0.5 * (I(y+2, x+1) - I(y, x+1)), i.e. a central
//          difference ALONG Y despite the filler's name), summed over the image channels with weight -0.33, exp();
//   gI_h = the same with EdgeY (filler.hpp:291-316: 0.5 * (I(y+1, x+2) - I(y+1, x)), along X);
//   dx = exp(-0.33 * sum_c gI_v) * (EdgeX * D),  dy = exp(-0.33 * sum_c gI_h) * (EdgeY * D)       (Eltwise PROD)
//   loss = weight * (sum|dx| + sum|dy|) / B                    (AbsLoss against zeros, abs_loss_layer.cu:10-26, loss_weight 10)
//   backward through AbsLoss: d|v|/dv = (v >= 0) - (v < 0)     (abs_loss_layer.cu:28-34: zero maps to ONE side, not to 0)
// The convolutions have no padding, so dx / dy live on the (H-2) x (W-2) interior.  Only the inverse depth receives a
// gradient (the image branch has lr_mult 0 and no trainable input).  `in_scale` multiplies the image as it is read
// (norm_imR2 = 0.004 * image, train.prototxt:124-171).
// No Caffe runtime exists in the build image: parity is UNPINNED; tests/ check this kernel pair against the oracle's
// restatement and the oracle against finite differences.
#include "dvf_common.h"

namespace {

constexpr int EX = 64, EY = 4;

// One thread per interior output (y, x) of image n: the two edge weights and the two inverse-depth differences.
struct EdgeVals { float wv, wh, dv, dh; };

__device__ __forceinline__ EdgeVals edge_vals(const float *__restrict__ D, const float *__restrict__ img, int C, int H, int W,
                                              int y, int x, float in_scale, float k) {
    // taps around the centre (y+1, x+1)
    const int up = y * W + x + 1, dn = (y + 2) * W + x + 1, lf = (y + 1) * W + x, rt = (y + 1) * W + x + 2;
    float sv = 0.f, sh = 0.f;
    for (int c = 0; c < C; ++c) {
        const float *p = img + (int64_t)c * H * W;
        const float a = __fmul_rn(in_scale, p[dn]), b = __fmul_rn(in_scale, p[up]);
        const float cc = __fmul_rn(in_scale, p[rt]), d = __fmul_rn(in_scale, p[lf]);
        sv += fabsf(0.5f * a - 0.5f * b);
        sh += fabsf(0.5f * cc - 0.5f * d);
    }
    EdgeVals e;
    e.wv = expf(-k * sv);
    e.wh = expf(-k * sh);
    e.dv = 0.5f * D[dn] - 0.5f * D[up];
    e.dh = 0.5f * D[rt] - 0.5f * D[lf];
    return e;
}

__global__ __launch_bounds__(256) void edge_smooth_fwd_kernel(const float *inv_depth, const float *img, float *partials, int C,
                                                              int H, int W, float in_scale, float k) {
    __shared__ float red[EY];
    const int n = blockIdx.z, x = blockIdx.x * EX + threadIdx.x, y = blockIdx.y * EY + threadIdx.y;
    float s = 0.f;
    if (x < W - 2 && y < H - 2) {
        const EdgeVals e = edge_vals(inv_depth + (int64_t)n * H * W, img + (int64_t)n * C * H * W, C, H, W, y, x, in_scale, k);
        s = fabsf(e.wv * e.dv) + fabsf(e.wh * e.dh);
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) red[threadIdx.y] = s;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[blk] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

__global__ __launch_bounds__(256) void edge_smooth_reduce_kernel(const float *partials, int64_t nblk, float scale, float *loss_out,
                                                                 int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < nblk; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
        loss_out[0] = accumulate ? loss_out[0] + t : t;
    }
}

// Gather-form backward: pixel (p, q) of D collects the four outputs whose stencils touch it.
__global__ __launch_bounds__(256) void edge_smooth_bwd_kernel(const float *inv_depth, const float *img, const float *grad_loss,
                                                              float *g_inv_depth, int C, int H, int W, float in_scale, float k,
                                                              float scale) {
    const int n = blockIdx.z, q = blockIdx.x * EX + threadIdx.x, p = blockIdx.y * EY + threadIdx.y;
    if (q >= W || p >= H) return;
    const float *D = inv_depth + (int64_t)n * H * W, *I = img + (int64_t)n * C * H * W;
    const float gl = grad_loss[0] * scale;
    auto caffe_sign = [](float v) { return v >= 0.f ? 1.f : -1.f; };     // abs_loss_layer.cu:31
    float g = 0.f;
    // vertical term: output (y, x) reads D(y+2, x+1) with +0.5 and D(y, x+1) with -0.5
    if (q >= 1 && q - 1 < W - 2) {
        if (p >= 2 && p - 2 < H - 2) { const EdgeVals e = edge_vals(D, I, C, H, W, p - 2, q - 1, in_scale, k); g += 0.5f * e.wv * caffe_sign(e.wv * e.dv); }
        if (p < H - 2) { const EdgeVals e = edge_vals(D, I, C, H, W, p, q - 1, in_scale, k); g -= 0.5f * e.wv * caffe_sign(e.wv * e.dv); }
    }
    // horizontal term: output (y, x) reads D(y+1, x+2) with +0.5 and D(y+1, x) with -0.5
    if (p >= 1 && p - 1 < H - 2) {
        if (q >= 2 && q - 2 < W - 2) { const EdgeVals e = edge_vals(D, I, C, H, W, p - 1, q - 2, in_scale, k); g += 0.5f * e.wh * caffe_sign(e.wh * e.dh); }
        if (q < W - 2) { const EdgeVals e = edge_vals(D, I, C, H, W, p - 1, q, in_scale, k); g -= 0.5f * e.wh * caffe_sign(e.wh * e.dh); }
    }
    g_inv_depth[(int64_t)n * H * W + (int64_t)p * W + q] = gl * g;
}

inline dim3 edge_grid(int B, int H, int W) { return dim3((W + EX - 1) / EX, (H + EY - 1) / EY, B); }

}  // namespace

extern "C" {

int64_t dvf_edge_smooth_partials_floats(int B, int H, int W) {
    const dim3 g = edge_grid(B, H, W);
    return (int64_t)g.x * g.y * g.z;
}

int dvf_edge_smooth_fwd(const float *inv_depth, const float *img, float *loss_out, float *partials, int B, int C, int H, int W,
                        float in_scale, float edge_k, float weight, int accumulate, void *stream) {
    if (!inv_depth || !img || !loss_out || !partials || B <= 0 || C <= 0 || H < 3 || W < 3 || B > 65535 ||
        (int64_t)C * H * W >= ((int64_t)1 << 31))
        return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    const dim3 grid = edge_grid(B, H, W);
    edge_smooth_fwd_kernel<<<grid, dim3(EX, EY), 0, st>>>(inv_depth, img, partials, C, H, W, in_scale, edge_k);
    DVF_LAUNCH_CHECK();
    edge_smooth_reduce_kernel<<<1, 256, 0, st>>>(partials, (int64_t)grid.x * grid.y * grid.z, weight / (float)B, loss_out, accumulate);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_edge_smooth_bwd(const float *inv_depth, const float *img, const float *grad_loss, float *g_inv_depth, int B, int C, int H,
                        int W, float in_scale, float edge_k, float weight, void *stream) {
    if (!inv_depth || !img || !grad_loss || !g_inv_depth || B <= 0 || C <= 0 || H < 3 || W < 3 || B > 65535 ||
        (int64_t)C * H * W >= ((int64_t)1 << 31))
        return DVF_ERR_INVALID_ARG;
    edge_smooth_bwd_kernel<<<edge_grid(B, H, W), dim3(EX, EY), 0, dvf_stream(stream)>>>(inv_depth, img, grad_loss, g_inv_depth, C, H,
                                                                                      W, in_scale, edge_k, weight / (float)B);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

}  // extern "C"
