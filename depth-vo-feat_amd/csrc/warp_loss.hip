// inverse_warp (image out), the smoothness loss and the stand-alone geometry helpers (the fused warp + photometric
// loss lives in photo_loss.hip).
//
// One launch per pyramid scale handles every reference view of a target pixel: depth and target are
// read once, the pixel->cam->SE3->pixel chain is evaluated in registers, the 2x2 source neighbourhood
// is gathered (wave = 64 consecutive pixels of one image row, so near-identity warps read near-contiguous
// lines), the exact-zero mask / explainability mask / |.| are applied and the partial sums are reduced
// wavefront-first.  The backward kernel recomputes the chain (SURVEY.md section 8d byte model) and
// produces grad depth, the [R|t] gradient partials (-> pose gradient in a finalize kernel), and on
// request grad target / grad source (scatter-add) / grad mask.
//
// Arithmetic follows the reference op by op (file:line in each helper) in fp32; products/sums whose
// rounding decides an exact comparison in the reference are written with __f*_rn so hipcc does not
// contract them into FMAs.
#include "warp_common.h"
namespace {

using namespace dvfw;

// ------------------------------------------------------------------ inverse_warp with an image output
struct WarpArgs {
    const float *img, *depth, *pose, *K, *Kinv, *grad_out;
    float *out, *g_img, *g_depth, *pose_ws;
    int B, C, H, W;
    uint32_t quat;
};

template <bool BORDER, bool ALIGN, bool PIX, bool BWD>
__global__ __launch_bounds__(256) void warp_kernel(WarpArgs a) {
    __shared__ ViewGeo geo;
    __shared__ float kinv[9], kmat[9];
    __shared__ float red[TY][DVF_MAX_VIEWS * 12];
    const int b = blockIdx.z, tid = threadIdx.y * TX + threadIdx.x;
    if (tid == 0) build_view(a.pose + (int64_t)b * 6, a.K + (int64_t)b * 9, (int)a.quat, &geo);
    if (tid >= 64 && tid < 73) kinv[tid - 64] = a.Kinv[(int64_t)b * 9 + tid - 64];
    if (tid >= 128 && tid < 137) kmat[tid - 128] = a.K[(int64_t)b * 9 + tid - 128];
    __syncthreads();
    const int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
    const int W = a.W, H = a.H, C = a.C;
    const int64_t HW = (int64_t)H * W;
    float pacc[1][12];
#pragma unroll
    for (int k = 0; k < 12; ++k) pacc[0][k] = 0.f;
    if (x < W && y < H) {
        const int64_t pix = (int64_t)y * W + x;
        const float d = a.depth[(int64_t)b * HW + pix];
        const float u = (float)x, v = (float)y;
        const float c0x = kinv[0] * u + kinv[1] * v + kinv[2];
        const float c0y = kinv[3] * u + kinv[4] * v + kinv[5];
        const float c0z = kinv[6] * u + kinv[7] * v + kinv[8];
        const float cx = c0x * d, cy = c0y * d, cz = c0z * d;
        const Samp s = project<BORDER, ALIGN, PIX>(geo, cx, cy, cz, W, H);
        const float *sp = a.img + (int64_t)b * C * HW;
        if (!BWD) {
            float *op = a.out + (int64_t)b * C * HW + pix;
            const TapAddr ta = tap_addr(s, W, H);
            for (int c = 0; c < C; ++c) op[c * HW] = blend(gather(sp + c * HW, ta), s);
        } else {
            const float *go = a.grad_out + (int64_t)b * C * HW + pix;
            float gix = 0.f, giy = 0.f;
            const bool xin0 = (unsigned)s.x0 < (unsigned)W, xin1 = (unsigned)(s.x0 + 1) < (unsigned)W;
            const bool yin0 = (unsigned)s.y0 < (unsigned)H, yin1 = (unsigned)(s.y0 + 1) < (unsigned)H;
            const TapAddr ta = tap_addr(s, W, H);
            for (int c = 0; c < C; ++c) {
                const Taps t = gather(sp + c * HW, ta);
                const float g = go[c * HW];
                float dox, doy;
                blend_grad(t, s, dox, doy);
                gix += g * dox;
                giy += g * doy;
                if (a.g_img) {
                    float *gp = a.g_img + ((int64_t)b * C + c) * HW + (int64_t)s.y0 * W + s.x0;
                    if (xin0 && yin0) atomicAdd(gp, g * s.wnw);
                    if (xin1 && yin0) atomicAdd(gp + 1, g * s.wne);
                    if (xin0 && yin1) atomicAdd(gp + W, g * s.wsw);
                    if (xin1 && yin1) atomicAdd(gp + W + 1, g * s.wse);
                }
            }
            const float gxq = gix * s.dix, gyq = giy * s.diy;
            const float gpx = gxq / s.Z, gpy = gyq / s.Z;
            const float gpz = s.zpass ? -(gxq * s.xq + gyq * s.yq) / s.Z : 0.f;
            const float gcx = geo.A[0] * gpx + geo.A[3] * gpy + geo.A[6] * gpz;
            const float gcy = geo.A[1] * gpx + geo.A[4] * gpy + geo.A[7] * gpz;
            const float gcz = geo.A[2] * gpx + geo.A[5] * gpy + geo.A[8] * gpz;
            if (a.g_depth) a.g_depth[(int64_t)b * HW + pix] = gcx * c0x + gcy * c0y + gcz * c0z;
            const float gyx = kmat[0] * gpx + kmat[3] * gpy + kmat[6] * gpz;
            const float gyy = kmat[1] * gpx + kmat[4] * gpy + kmat[7] * gpz;
            const float gyz = kmat[2] * gpx + kmat[5] * gpy + kmat[8] * gpz;
            pacc[0][0] = gyx; pacc[0][1] = gyy; pacc[0][2] = gyz;
            pacc[0][3] = gyx * cx; pacc[0][4] = gyx * cy; pacc[0][5] = gyx * cz;
            pacc[0][6] = gyy * cx; pacc[0][7] = gyy * cy; pacc[0][8] = gyy * cz;
            pacc[0][9] = gyz * cx; pacc[0][10] = gyz * cy; pacc[0][11] = gyz * cz;
        }
    }
    if (BWD && a.pose_ws) reduce_pose_partials<1>(pacc, 1, a.B, b, a.pose_ws, red);
}

// ------------------------------------------------------------------ smoothness loss
constexpr int SX = 64, SY = 4, HALO = 2;

// Second differences with the reference's association (loss_functions.py:24-38): first differences are
// rounded once, then differenced.
struct SmTile {
    float v[SY + 2 * HALO][SX + 2 * HALO];
};

__device__ __forceinline__ void load_tile(const float *plane, int H, int W, int bx, int by, SmTile &t) {
    const int tid = threadIdx.y * SX + threadIdx.x;
    for (int i = tid; i < (SY + 2 * HALO) * (SX + 2 * HALO); i += SX * SY) {
        const int ty = i / (SX + 2 * HALO), tx = i % (SX + 2 * HALO);
        const int gy = by + ty - HALO, gx = bx + tx - HALO;
        t.v[ty][tx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? plane[(int64_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
}

__device__ __forceinline__ float d_dx2(const SmTile &t, int ty, int tx) {   // origin (y, x): needs x+2
    return __fsub_rn(__fsub_rn(t.v[ty][tx + 2], t.v[ty][tx + 1]), __fsub_rn(t.v[ty][tx + 1], t.v[ty][tx]));
}
__device__ __forceinline__ float d_dy2(const SmTile &t, int ty, int tx) {
    return __fsub_rn(__fsub_rn(t.v[ty + 2][tx], t.v[ty + 1][tx]), __fsub_rn(t.v[ty + 1][tx], t.v[ty][tx]));
}
__device__ __forceinline__ float d_dxdy(const SmTile &t, int ty, int tx) {  // D_dy of dx
    return __fsub_rn(__fsub_rn(t.v[ty + 1][tx + 1], t.v[ty + 1][tx]), __fsub_rn(t.v[ty][tx + 1], t.v[ty][tx]));
}
__device__ __forceinline__ float d_dydx(const SmTile &t, int ty, int tx) {  // D_dx of dy
    return __fsub_rn(__fsub_rn(t.v[ty + 1][tx + 1], t.v[ty][tx + 1]), __fsub_rn(t.v[ty + 1][tx], t.v[ty][tx]));
}

__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float *map, float *partials, int H, int W) {
    __shared__ SmTile t;
    __shared__ float red[SY][4];
    const int n = blockIdx.z, bx = blockIdx.x * SX, by = blockIdx.y * SY;
    load_tile(map + (int64_t)n * H * W, H, W, bx, by, t);
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    const int tx = threadIdx.x + HALO, ty = threadIdx.y + HALO;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (x < W && y < H) {
        if (x + 2 < W) s[0] = fabsf(d_dx2(t, ty, tx));
        if (x + 1 < W && y + 1 < H) { s[1] = fabsf(d_dxdy(t, ty, tx)); s[2] = fabsf(d_dydx(t, ty, tx)); }
        if (y + 2 < H) s[3] = fabsf(d_dy2(t, ty, tx));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float r = wave_sum(s[k]);
        if (threadIdx.x == 0) red[threadIdx.y][k] = r;
    }
    __syncthreads();
    const int tid = threadIdx.y * SX + threadIdx.x;
    if (tid < 4) {
        const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[blk * 4 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

__global__ __launch_bounds__(256) void smooth_reduce_kernel(const float *partials, int64_t nblk, float w0, float w1,
                                                            float w2, float w3, float *loss_out, int accumulate) {
    __shared__ float red[4];
    const float wk[4] = {w0, w1, w2, w3};
    float total = 0.f;
    for (int k = 0; k < 4; ++k) {
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < nblk; i += 256) s += partials[i * 4 + k];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        total += ((red[0] + red[1]) + (red[2] + red[3])) * wk[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = accumulate ? loss_out[0] + total : total;
}

// Gather-form backward: every pixel sums the signs of the stencils that touch it.
__global__ __launch_bounds__(256) void smooth_bwd_kernel(const float *map, const float *grad_loss, float *g_map, int H,
                                                         int W, float w1, float w2, float w3) {
    __shared__ SmTile t;
    const int n = blockIdx.z, bx = blockIdx.x * SX, by = blockIdx.y * SY;
    load_tile(map + (int64_t)n * H * W, H, W, bx, by, t);
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    if (x >= W || y >= H) return;
    const int tx = threadIdx.x + HALO, ty = threadIdx.y + HALO;
    const float gl = grad_loss[0];
    // dx2 with origin (y, xo), xo in {x-2, x-1, x}: coefficients +1, -2, +1 on (xo, xo+1, xo+2)
    float g1 = 0.f;
    if (x - 2 >= 0) g1 += sgn(d_dx2(t, ty, tx - 2));                       // x is xo+2
    if (x - 1 >= 0 && x + 1 < W) g1 -= 2.f * sgn(d_dx2(t, ty, tx - 1));    // x is xo+1
    if (x + 2 < W) g1 += sgn(d_dx2(t, ty, tx));                            // x is xo
    float g3 = 0.f;
    if (y - 2 >= 0) g3 += sgn(d_dy2(t, ty - 2, tx));
    if (y - 1 >= 0 && y + 1 < H) g3 -= 2.f * sgn(d_dy2(t, ty - 1, tx));
    if (y + 2 < H) g3 += sgn(d_dy2(t, ty, tx));
    // mixed terms with origin (yo, xo): +1 on (yo+1,xo+1) and (yo,xo), -1 on (yo+1,xo) and (yo,xo+1)
    float g2 = 0.f;
    if (x + 1 < W && y + 1 < H) g2 += sgn(d_dxdy(t, ty, tx)) + sgn(d_dydx(t, ty, tx));                 // (yo,xo)
    if (x - 1 >= 0 && y + 1 < H) g2 -= sgn(d_dxdy(t, ty, tx - 1)) + sgn(d_dydx(t, ty, tx - 1));         // (yo,xo+1)
    if (x + 1 < W && y - 1 >= 0) g2 -= sgn(d_dxdy(t, ty - 1, tx)) + sgn(d_dydx(t, ty - 1, tx));         // (yo+1,xo)
    if (x - 1 >= 0 && y - 1 >= 0) g2 += sgn(d_dxdy(t, ty - 1, tx - 1)) + sgn(d_dydx(t, ty - 1, tx - 1)); // (yo+1,xo+1)
    g_map[(int64_t)n * H * W + (int64_t)y * W + x] = gl * (g1 * w1 + g2 * w2 + g3 * w3);
}

// ------------------------------------------------------------------ stand-alone geometry helpers (inverse_warp.py API)
__global__ void pose_vec2mat_kernel(const float *pose, float *out, int n, uint32_t quat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float R[9];
    pose_to_R(pose + (int64_t)i * 6, (int)quat, R);
    float *o = out + (int64_t)i * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[r * 4 + 0] = R[r * 3 + 0]; o[r * 4 + 1] = R[r * 3 + 1]; o[r * 4 + 2] = R[r * 3 + 2];
        const float *p = pose + (int64_t)i * 6;
        o[r * 4 + 3] = quat == 2 ? R[r * 3 + 0] * p[3] + R[r * 3 + 1] * p[4] + R[r * 3 + 2] * p[5] : p[r];   // se3: t = R u
    }
}

// g_mat [n,3,4] -> workspace layout of pose_finalize_kernel (g_t[3], g_R[9])
__global__ void mat_grad_to_ws_kernel(const float *g_mat, float *ws, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *g = g_mat + (int64_t)i * 12;
    float *w = ws + (int64_t)i * 12;
    w[0] = g[3]; w[1] = g[7]; w[2] = g[11];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) w[3 + r * 3 + c] = g[r * 4 + c];
}

// cam = (Kinv @ (u, v, 1)) * depth   (inverse_warp.py:26-40); BWD: g_depth = sum_c g_cam_c * (Kinv @ (u,v,1))_c
template <bool BWD>
__global__ void pixel2cam_kernel(const float *depth, const float *Kinv, const float *g_cam, float *out, int H, int W,
                                 int64_t total) {
    const int64_t HW = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int64_t pix = i - (int64_t)b * HW;
        const float u = (float)(pix % W), v = (float)(pix / W);
        const float *k = Kinv + (int64_t)b * 9;
        const float c0 = k[0] * u + k[1] * v + k[2], c1 = k[3] * u + k[4] * v + k[5], c2 = k[6] * u + k[7] * v + k[8];
        if (!BWD) {
            const float d = depth[i];
            float *o = out + (int64_t)b * 3 * HW + pix;
            o[0] = c0 * d; o[HW] = c1 * d; o[2 * HW] = c2 * d;
        } else {
            const float *g = g_cam + (int64_t)b * 3 * HW + pix;
            out[i] = g[0] * c0 + g[HW] * c1 + g[2 * HW] * c2;
        }
    }
}

// cam [B,3,H,W] -> normalised grid [B,H,W,2]   (inverse_warp.py:43-74)
__global__ void cam2pixel_fwd_kernel(const float *cam, const float *rot, const float *tr, float *grid, int H, int W,
                                     int zeros_pad, int64_t total) {
    const int64_t HW = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int64_t pix = i - (int64_t)b * HW;
        const float *c = cam + (int64_t)b * 3 * HW + pix;
        float p[3] = {c[0], c[HW], c[2 * HW]};
        if (rot) {
            const float *r = rot + (int64_t)b * 9;
            const float x = p[0], y = p[1], z = p[2];
            p[0] = r[0] * x + r[1] * y + r[2] * z; p[1] = r[3] * x + r[4] * y + r[5] * z; p[2] = r[6] * x + r[7] * y + r[8] * z;
        }
        if (tr) { p[0] += tr[b * 3 + 0]; p[1] += tr[b * 3 + 1]; p[2] += tr[b * 3 + 2]; }
        const float Z = fmaxf(p[2], 1e-3f);
        float xn = __fsub_rn(__fdiv_rn(2.f * (p[0] / Z), (float)(W - 1)), 1.f);
        float yn = __fsub_rn(__fdiv_rn(2.f * (p[1] / Z), (float)(H - 1)), 1.f);
        if (zeros_pad) {
            if (xn > 1.f || xn < -1.f) xn = 2.f;
            if (yn > 1.f || yn < -1.f) yn = 2.f;
        }
        grid[2 * i] = xn;
        grid[2 * i + 1] = yn;
    }
}

// backward: g_cam per pixel; g_rot (9) and g_tr (3) per batch element through block partials + atomics (ws [B,12])
__global__ __launch_bounds__(256) void cam2pixel_bwd_kernel(const float *cam, const float *rot, const float *tr,
                                                            const float *g_grid, float *g_cam, float *ws, int H, int W,
                                                            int zeros_pad, int chunks) {
    __shared__ float red[4][12];
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const int HW = H * W, per = (HW + chunks - 1) / chunks, beg = chunk * per, end = min(HW, beg + per);
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int pix = beg + threadIdx.x; pix < end; pix += 256) {
        const float *c = cam + (int64_t)b * 3 * HW + pix;
        const float x = c[0], y = c[HW], z = c[2 * HW];
        float p[3] = {x, y, z};
        float r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (rot) {
#pragma unroll
            for (int k = 0; k < 9; ++k) r[k] = rot[(int64_t)b * 9 + k];
            p[0] = r[0] * x + r[1] * y + r[2] * z; p[1] = r[3] * x + r[4] * y + r[5] * z; p[2] = r[6] * x + r[7] * y + r[8] * z;
        }
        if (tr) { p[0] += tr[b * 3 + 0]; p[1] += tr[b * 3 + 1]; p[2] += tr[b * 3 + 2]; }
        const bool zpass = p[2] >= 1e-3f;
        const float Z = fmaxf(p[2], 1e-3f), xq = p[0] / Z, yq = p[1] / Z;
        const float xn = __fsub_rn(__fdiv_rn(2.f * xq, (float)(W - 1)), 1.f), yn = __fsub_rn(__fdiv_rn(2.f * yq, (float)(H - 1)), 1.f);
        float gx = g_grid[2 * ((int64_t)b * HW + pix)], gy = g_grid[2 * ((int64_t)b * HW + pix) + 1];
        if (zeros_pad) {
            if (xn > 1.f || xn < -1.f) gx = 0.f;
            if (yn > 1.f || yn < -1.f) gy = 0.f;
        }
        const float gxq = gx * 2.f / (float)(W - 1), gyq = gy * 2.f / (float)(H - 1);
        const float gp0 = gxq / Z, gp1 = gyq / Z, gp2 = zpass ? -(gxq * xq + gyq * yq) / Z : 0.f;
        if (g_cam) {
            float *g = g_cam + (int64_t)b * 3 * HW + pix;
            g[0] = r[0] * gp0 + r[3] * gp1 + r[6] * gp2;
            g[HW] = r[1] * gp0 + r[4] * gp1 + r[7] * gp2;
            g[2 * HW] = r[2] * gp0 + r[5] * gp1 + r[8] * gp2;
        }
        acc[0] += gp0; acc[1] += gp1; acc[2] += gp2;
        acc[3] += gp0 * x; acc[4] += gp0 * y; acc[5] += gp0 * z;
        acc[6] += gp1 * x; acc[7] += gp1 * y; acc[8] += gp1 * z;
        acc[9] += gp2 * x; acc[10] += gp2 * y; acc[11] += gp2 * z;
    }
    if (!ws) return;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float s = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12)
        atomicAdd(&ws[(int64_t)b * 12 + threadIdx.x],
                  (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

inline dim3 pix_grid(int B, int H, int W) { return dim3((W + TX - 1) / TX, (H + TY - 1) / TY, B); }

}  // namespace

extern "C" {

int64_t dvf_pose_ws_floats(int V, int B) { return (int64_t)V * B * 12; }

int64_t dvf_smooth_partials_floats(int N, int H, int W) {
    return (int64_t)((W + SX - 1) / SX) * ((H + SY - 1) / SY) * N * 4;
}

int dvf_inverse_warp_fwd(const float *img, const float *depth, const float *pose, const float *K, const float *Kinv,
                         float *out, int B, int C, int H, int W, uint32_t flags, void *stream) {
    if (!img || !depth || !pose || !K || !Kinv || !out || B <= 0 || C <= 0 || H < 2 || W < 2 || B > 65535)
        return DVF_ERR_INVALID_ARG;
    WarpArgs a{img, depth, pose, K, Kinv, nullptr, out, nullptr, nullptr, nullptr, B, C, H, W, rot_mode(flags)};
    return dispatch_mode(flags, [&](auto border, auto align, auto pix) {
        warp_kernel<decltype(border)::value, decltype(align)::value, decltype(pix)::value, false>
            <<<pix_grid(B, H, W), dim3(TX, TY), 0, dvf_stream(stream)>>>(a);
        DVF_LAUNCH_CHECK();
        return DVF_OK;
    });
}

int dvf_inverse_warp_bwd(const float *img, const float *depth, const float *pose, const float *K, const float *Kinv,
                         const float *grad_out, float *g_img, float *g_depth, float *g_pose, float *pose_ws, int B,
                         int C, int H, int W, uint32_t flags, void *stream) {
    if (!img || !depth || !pose || !K || !Kinv || !grad_out || B <= 0 || C <= 0 || H < 2 || W < 2 || B > 65535)
        return DVF_ERR_INVALID_ARG;
    if (g_pose && !pose_ws) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    if (g_pose && hipMemsetAsync(pose_ws, 0, sizeof(float) * 12 * B, st) != hipSuccess) return DVF_ERR_LAUNCH;
    WarpArgs a{img, depth, pose, K, Kinv, grad_out, nullptr, g_img, g_depth, g_pose ? pose_ws : nullptr,
               B, C, H, W, rot_mode(flags)};
    const int rc = dispatch_mode(flags, [&](auto border, auto align, auto pix) {
        warp_kernel<decltype(border)::value, decltype(align)::value, decltype(pix)::value, true>
            <<<pix_grid(B, H, W), dim3(TX, TY), 0, st>>>(a);
        DVF_LAUNCH_CHECK();
        return DVF_OK;
    });
    if (rc != DVF_OK) return rc;
    if (g_pose) {
        pose_finalize_kernel<<<(B + 63) / 64, 64, 0, st>>>(pose, pose_ws, g_pose, B, rot_mode(flags));
        DVF_LAUNCH_CHECK();
    }
    return DVF_OK;
}

int dvf_pose_vec2mat_fwd(const float *pose, float *out, int n, uint32_t flags, void *stream) {
    if (!pose || !out || n <= 0) return DVF_ERR_INVALID_ARG;
    pose_vec2mat_kernel<<<(n + 63) / 64, 64, 0, dvf_stream(stream)>>>(pose, out, n, rot_mode(flags));
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_pose_vec2mat_bwd(const float *pose, const float *g_mat, float *g_pose, float *ws, int n, uint32_t flags,
                         void *stream) {
    if (!pose || !g_mat || !g_pose || !ws || n <= 0) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    mat_grad_to_ws_kernel<<<(n + 63) / 64, 64, 0, st>>>(g_mat, ws, n);
    DVF_LAUNCH_CHECK();
    pose_finalize_kernel<<<(n + 63) / 64, 64, 0, st>>>(pose, ws, g_pose, n, rot_mode(flags));
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_pixel2cam_fwd(const float *depth, const float *Kinv, float *cam, int B, int H, int W, void *stream) {
    if (!depth || !Kinv || !cam || B <= 0 || H <= 0 || W <= 0) return DVF_ERR_INVALID_ARG;
    const int64_t total = (int64_t)B * H * W;
    const int nb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    pixel2cam_kernel<false><<<nb, 256, 0, dvf_stream(stream)>>>(depth, Kinv, nullptr, cam, H, W, total);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_pixel2cam_bwd(const float *Kinv, const float *g_cam, float *g_depth, int B, int H, int W, void *stream) {
    if (!Kinv || !g_cam || !g_depth || B <= 0 || H <= 0 || W <= 0) return DVF_ERR_INVALID_ARG;
    const int64_t total = (int64_t)B * H * W;
    const int nb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    pixel2cam_kernel<true><<<nb, 256, 0, dvf_stream(stream)>>>(nullptr, Kinv, g_cam, g_depth, H, W, total);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_cam2pixel_fwd(const float *cam, const float *rot, const float *tr, float *grid, int B, int H, int W,
                      uint32_t flags, void *stream) {
    if (!cam || !grid || B <= 0 || H < 2 || W < 2) return DVF_ERR_INVALID_ARG;
    const int64_t total = (int64_t)B * H * W;
    const int nb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    cam2pixel_fwd_kernel<<<nb, 256, 0, dvf_stream(stream)>>>(cam, rot, tr, grid, H, W, (flags & DVF_PAD_BORDER) ? 0 : 1, total);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_cam2pixel_bwd(const float *cam, const float *rot, const float *tr, const float *g_grid, float *g_cam,
                      float *g_rot_tr_ws, int B, int H, int W, uint32_t flags, void *stream) {
    if (!cam || !g_grid || B <= 0 || H < 2 || W < 2 || B > 4096) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    if (g_rot_tr_ws && hipMemsetAsync(g_rot_tr_ws, 0, sizeof(float) * 12 * B, st) != hipSuccess) return DVF_ERR_LAUNCH;
    int chunks = (H * W + 8191) / 8192;
    if (chunks < 1) chunks = 1;
    cam2pixel_bwd_kernel<<<B * chunks, 256, 0, st>>>(cam, rot, tr, g_grid, g_cam, g_rot_tr_ws, H, W,
                                                     (flags & DVF_PAD_BORDER) ? 0 : 1, chunks);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

static void smooth_weights(int N, int H, int W, float weight, float *w) {
    // mean over N*H*(W-2), N*(H-1)*(W-1) (twice), N*(H-2)*W elements     loss_functions.py:39
    w[0] = (W > 2) ? weight / ((float)N * (float)H * (float)(W - 2)) : 0.f;
    w[1] = (W > 1 && H > 1) ? weight / ((float)N * (float)(H - 1) * (float)(W - 1)) : 0.f;
    w[2] = (H > 2) ? weight / ((float)N * (float)(H - 2) * (float)W) : 0.f;
}

int dvf_smooth_loss_fwd(const float *map, float *loss_out, float *partials, int N, int H, int W, float weight,
                        int accumulate, void *stream) {
    if (!map || !loss_out || !partials || N <= 0 || H < 3 || W < 3 || N > 65535) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    const dim3 grid((W + SX - 1) / SX, (H + SY - 1) / SY, N);
    smooth_fwd_kernel<<<grid, dim3(SX, SY), 0, st>>>(map, partials, H, W);
    DVF_LAUNCH_CHECK();
    float w[3];
    smooth_weights(N, H, W, weight, w);
    smooth_reduce_kernel<<<1, 256, 0, st>>>(partials, (int64_t)grid.x * grid.y * grid.z, w[0], w[1], w[1], w[2],
                                            loss_out, accumulate);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_smooth_loss_bwd(const float *map, const float *grad_loss, float *g_map, int N, int H, int W, float weight,
                        void *stream) {
    if (!map || !grad_loss || !g_map || N <= 0 || H < 3 || W < 3 || N > 65535) return DVF_ERR_INVALID_ARG;
    float w[3];
    smooth_weights(N, H, W, weight, w);
    const dim3 grid((W + SX - 1) / SX, (H + SY - 1) / SY, N);
    smooth_bwd_kernel<<<grid, dim3(SX, SY), 0, dvf_stream(stream)>>>(map, grad_loss, g_map, H, W, w[0], w[1], w[2]);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

}  // extern "C"
