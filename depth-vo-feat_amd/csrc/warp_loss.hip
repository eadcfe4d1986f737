// Fused inverse-warp + photometric-L1 kernels, inverse_warp (image out) and the smoothness loss.
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
#include "dvf_common.h"

namespace {

constexpr int TX = 64;   // pixels along a row per wave
constexpr int TY = 4;    // rows per block pass (one wave each)
constexpr int RPT = 1;   // row passes per block (measured: 4 is slower -- fewer, longer blocks -- despite 4x fewer reductions)

struct ViewGeo {         // per (view, batch element); built once per block in LDS
    float A[9];          // K @ R           inverse_warp.py:188 (rotation part)
    float tr[3];         // K @ t           inverse_warp.py:188 (last column)
};

__device__ __forceinline__ void mat3mul(const float *a, const float *b, float *o) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

// pose -> R.  mode 0 euler (tx,ty,tz,rx,ry,rz): inverse_warp.py:77-114 (R = Rx @ Ry @ Rz); mode 1 quat: :117-138;
// mode 2 se3 (wx,wy,wz,ux,uy,uz): exponential map of se3_generate.py:13-43 / caffe/python/pygeometry.py:31-60
// (R = I + sin(th)/th [w]x + 2 sin^2(th/2)/th^2 [w]x^2, first-order for th^2 < 1e-12).
__device__ void pose_to_R(const float *p, int mode, float *R) {
    if (mode == 0) {
        const float cx = cosf(p[3]), sx = sinf(p[3]);
        const float cy = cosf(p[4]), sy = sinf(p[4]);
        const float cz = cosf(p[5]), sz = sinf(p[5]);
        const float X[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
        const float Y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
        const float Z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
        float XY[9];
        mat3mul(X, Y, XY);
        mat3mul(XY, Z, R);
    } else if (mode == 1) {
        const float n = sqrtf(1.f + p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
        const float w = 1.f / n, x = p[3] / n, y = p[4] / n, z = p[5] / n;
        const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
        const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
        R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
        R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;   R[5] = 2 * yz - 2 * wx;
        R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
    } else {
        const float wx = p[0], wy = p[1], wz = p[2];
        const float Wx[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        const float th2 = wx * wx + wy * wy + wz * wz;
        float c1 = 1.f, c2 = 0.f;
        if (th2 >= 1e-12f) {
            const float th = sqrtf(th2), sh = sinf(0.5f * th);
            c1 = sinf(th) / th;
            c2 = 2.f * sh * sh / th2;
        }
        float W2[9];
        mat3mul(Wx, Wx, W2);
#pragma unroll
        for (int e = 0; e < 9; ++e) R[e] = ((e % 4 == 0) ? 1.f : 0.f) + c1 * Wx[e] + c2 * W2[e];
    }
}

__device__ void build_view(const float *pose6, const float *K, int mode, ViewGeo *g) {
    float R[9], t[3];
    pose_to_R(pose6, mode, R);
    if (mode == 2) {                                     // se3: t = R u   (se3_generate.py:47)
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = R[i * 3 + 0] * pose6[3] + R[i * 3 + 1] * pose6[4] + R[i * 3 + 2] * pose6[5];
    } else {
        t[0] = pose6[0]; t[1] = pose6[1]; t[2] = pose6[2];
    }
    mat3mul(K, R, g->A);
#pragma unroll
    for (int i = 0; i < 3; ++i) g->tr[i] = K[i * 3 + 0] * t[0] + K[i * 3 + 1] * t[1] + K[i * 3 + 2] * t[2];
}

// Everything the sampler needs for one (pixel, view).
struct Samp {
    float ix, iy;            // un-normalised source coordinates
    float xq, yq, Z;         // X/Z, Y/Z, clamped Z
    float dix, diy;          // d ix / d xq and d iy / d yq (0 where the reference cuts the gradient)
    bool zpass;              // clamp(min=1e-3) passes gradient
    int x0, y0;
    float wnw, wne, wsw, wse;
};

template <bool BORDER, bool ALIGN, bool PIX = false>
__device__ __forceinline__ Samp project(const ViewGeo &g, float cx, float cy, float cz, int W, int H) {
    Samp s;
    // p = (K R) cam + K t                                   inverse_warp.py:55-60
    const float px = g.A[0] * cx + g.A[1] * cy + g.A[2] * cz + g.tr[0];
    const float py = g.A[3] * cx + g.A[4] * cy + g.A[5] * cz + g.tr[1];
    const float pz = g.A[6] * cx + g.A[7] * cy + g.A[8] * cz + g.tr[2];
    if (PIX) {
        // pixel-coordinate front end (Caffe PinHole + InverseWarping semantics, pin_hole_layer.cu:10-50,
        // inverse_warping_layer.cu:10-52): u = fx X / (Z + 1e-12) + cx sampled directly, each tap bounds-checked
        s.zpass = true;
        s.Z = pz + 1e-12f;
        s.xq = px / s.Z;
        s.yq = py / s.Z;
        s.ix = s.xq;
        s.iy = s.yq;
        s.dix = 1.f;
        s.diy = 1.f;
        const float fx = floorf(s.ix), fy = floorf(s.iy);
        s.x0 = (fx >= -2.f && fx <= (float)W + 1.f) ? (int)fx : -4;
        s.y0 = (fy >= -2.f && fy <= (float)H + 1.f) ? (int)fy : -4;
        const float ex = (fx + 1.f) - s.ix, ey = (fy + 1.f) - s.iy, dx = s.ix - fx, dy = s.iy - fy;
        s.wnw = ex * ey; s.wne = dx * ey; s.wsw = ex * dy; s.wse = dx * dy;
        return s;
    }
    s.zpass = pz >= 1e-3f;
    s.Z = fmaxf(pz, 1e-3f);                                 // :63
    s.xq = px / s.Z;
    s.yq = py / s.Z;
    float xn = __fsub_rn(__fdiv_rn(2.f * s.xq, (float)(W - 1)), 1.f);   // :65
    float yn = __fsub_rn(__fdiv_rn(2.f * s.yq, (float)(H - 1)), 1.f);   // :66
    float mx = 2.f / (float)(W - 1), my = 2.f / (float)(H - 1);
    if (!BORDER) {                                          // :67-71 (overwrite with 2, gradient cut)
        if (xn > 1.f || xn < -1.f) { xn = 2.f; mx = 0.f; }
        if (yn > 1.f || yn < -1.f) { yn = 2.f; my = 0.f; }
    }
    // grid_sampler un-normalise (aten GridSampler.h): align_corners ? (x+1)/2*(size-1) : ((x+1)*size-1)/2
    if (ALIGN) {
        s.ix = __fmul_rn(__fmul_rn(__fadd_rn(xn, 1.f), 0.5f), (float)(W - 1));
        s.iy = __fmul_rn(__fmul_rn(__fadd_rn(yn, 1.f), 0.5f), (float)(H - 1));
        mx *= 0.5f * (float)(W - 1);
        my *= 0.5f * (float)(H - 1);
    } else {
        s.ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(xn, 1.f), (float)W), 1.f), 0.5f);
        s.iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(yn, 1.f), (float)H), 1.f), 0.5f);
        mx *= 0.5f * (float)W;
        my *= 0.5f * (float)H;
    }
    if (BORDER) {                                           // clip_coordinates_set_grad
        if (s.ix < 0.f) { s.ix = 0.f; mx = 0.f; } else if (s.ix > (float)(W - 1)) { s.ix = (float)(W - 1); mx = 0.f; }
        if (s.iy < 0.f) { s.iy = 0.f; my = 0.f; } else if (s.iy > (float)(H - 1)) { s.iy = (float)(H - 1); my = 0.f; }
    }
    s.dix = mx;
    s.diy = my;
    const float fx = floorf(s.ix), fy = floorf(s.iy);
    // NaN / huge coordinates: keep the integer conversion defined; such taps are out of bounds anyway
    s.x0 = (fx >= -2.f && fx <= (float)W + 1.f) ? (int)fx : -4;
    s.y0 = (fy >= -2.f && fy <= (float)H + 1.f) ? (int)fy : -4;
    const float ex = __fsub_rn(__fadd_rn(fx, 1.f), s.ix), ey = __fsub_rn(__fadd_rn(fy, 1.f), s.iy);   // ix_se - ix
    const float dx = __fsub_rn(s.ix, fx), dy = __fsub_rn(s.iy, fy);                                  // ix - ix_nw
    s.wnw = __fmul_rn(ex, ey);
    s.wne = __fmul_rn(dx, ey);
    s.wsw = __fmul_rn(ex, dy);
    s.wse = __fmul_rn(dx, dy);
    return s;
}

struct Taps { float nw, ne, sw, se; };

// Tap addressing of one (pixel, view), shared by all channels.  The two taps of a row are adjacent, so each row is ONE
// unconditional 8-byte load (4-byte aligned global_load_dwordx2) from a clamped, always valid pair position; the
// warp kernels are bound by the number of vector-memory instructions, not by bytes.
struct __attribute__((packed, aligned(4))) Pair { float x, y; };
struct TapAddr { int o_top, o_bot; bool straight; bool v_nw, v_ne, v_sw, v_se; };

__device__ __forceinline__ TapAddr tap_addr(const Samp &s, int W, int H) {
    TapAddr a;
    const bool xin0 = (unsigned)s.x0 < (unsigned)W, xin1 = (unsigned)(s.x0 + 1) < (unsigned)W;
    const bool yin0 = (unsigned)s.y0 < (unsigned)H, yin1 = (unsigned)(s.y0 + 1) < (unsigned)H;
    const int xb = min(max(s.x0, 0), W - 2);               // pair (xb, xb+1) is always inside the row
    const int y0 = min(max(s.y0, 0), H - 1), y1 = min(max(s.y0 + 1, 0), H - 1);
    a.o_top = y0 * W + xb;
    a.o_bot = y1 * W + xb;
    a.straight = (s.x0 == xb);                             // else the pair is shifted by one (x0 = -1 or W-1)
    a.v_nw = xin0 && yin0; a.v_ne = xin1 && yin0; a.v_sw = xin0 && yin1; a.v_se = xin1 && yin1;
    return a;
}

__device__ __forceinline__ Taps gather(const float *__restrict__ plane, const TapAddr &a) {
    Taps t;
    const Pair top = *reinterpret_cast<const Pair *>(plane + a.o_top);
    const Pair bot = *reinterpret_cast<const Pair *>(plane + a.o_bot);
    t.nw = a.v_nw ? (a.straight ? top.x : top.y) : 0.f;
    t.ne = a.v_ne ? (a.straight ? top.y : top.x) : 0.f;
    t.sw = a.v_sw ? (a.straight ? bot.x : bot.y) : 0.f;
    t.se = a.v_se ? (a.straight ? bot.y : bot.x) : 0.f;
    return t;
}

__device__ __forceinline__ float blend(const Taps &t, const Samp &s) {
    // aten accumulates nw, ne, sw, se in this order with separate roundings
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(t.nw, s.wnw), __fmul_rn(t.ne, s.wne)), __fmul_rn(t.sw, s.wsw)),
                     __fmul_rn(t.se, s.wse));
}

// d out / d ix and d out / d iy for one channel (aten grid_sampler_2d_backward)
__device__ __forceinline__ void blend_grad(const Taps &t, const Samp &s, float &dox, float &doy) {
    const float fx = floorf(s.ix), fy = floorf(s.iy);
    const float ex = (fx + 1.f) - s.ix, ey = (fy + 1.f) - s.iy, dx = s.ix - fx, dy = s.iy - fy;
    dox = -t.nw * ey + t.ne * ey - t.sw * dy + t.se * dy;
    doy = -t.nw * ex - t.ne * dx + t.sw * ex + t.se * dx;
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

struct PhotoArgs {
    const float *tgt;
    const float *src[DVF_MAX_VIEWS];
    const float *depth, *pose, *K, *Kinv, *mask;
    float *partials;
    // backward
    const float *grad_loss;
    float *g_depth, *g_tgt, *g_mask, *pose_ws;
    float *g_src[DVF_MAX_VIEWS];
    int B, C, H, W, V;
    uint32_t quat;
};

__device__ __forceinline__ void block_setup(const PhotoArgs &a, int b, int tid, ViewGeo *geo, float *kinv, float *kmat) {
    if (tid < a.V) build_view(a.pose + ((int64_t)tid * a.B + b) * 6, a.K + (int64_t)b * 9, (int)a.quat, &geo[tid]);
    if (tid >= 64 && tid < 73) kinv[tid - 64] = a.Kinv[(int64_t)b * 9 + tid - 64];
    if (kmat && tid >= 128 && tid < 137) kmat[tid - 128] = a.K[(int64_t)b * 9 + tid - 128];
    __syncthreads();
}

template <bool BORDER, bool ALIGN, bool PIX, int NV>
__global__ __launch_bounds__(256) void photo_fwd_kernel(PhotoArgs a) {
    __shared__ ViewGeo geo[DVF_MAX_VIEWS];
    __shared__ float kinv[9];
    __shared__ float red[TY][DVF_MAX_VIEWS];
    const int b = blockIdx.z, tid = threadIdx.y * TX + threadIdx.x;
    block_setup(a, b, tid, geo, kinv, nullptr);
    const int x = blockIdx.x * TX + threadIdx.x;
    const int W = a.W, H = a.H, C = a.C;
    const int64_t HW = (int64_t)H * W;
    float lsum[NV];
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) lsum[vi] = 0.f;
    for (int rp = 0; rp < RPT; ++rp) {
    const int y = (blockIdx.y * RPT + rp) * TY + threadIdx.y;
    if (x < W && y < H) {
        const int64_t pix = (int64_t)y * W + x;
        const float d = a.depth[(int64_t)b * HW + pix];
        // cam = (Kinv @ (u, v, 1)) * depth                  inverse_warp.py:38-40
        const float u = (float)x, v = (float)y;
        const float cx = (kinv[0] * u + kinv[1] * v + kinv[2]) * d;
        const float cy = (kinv[3] * u + kinv[4] * v + kinv[5]) * d;
        const float cz = (kinv[6] * u + kinv[7] * v + kinv[8]) * d;
        const float *tg = a.tgt + (int64_t)b * C * HW + pix;
        // all views are projected first, then every channel group gathers for ALL views at once: with V = 2 and
        // 4 channels that is 32 source loads + 4 target loads in flight per lane instead of 4
        Samp s[NV];
        TapAddr ta[NV];
        float acc[NV];
        bool nz[NV];
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) {
            s[vi] = project<BORDER, ALIGN, PIX>(geo[vi], cx, cy, cz, W, H);
            ta[vi] = tap_addr(s[vi], W, H);
            acc[vi] = 0.f;
            nz[vi] = false;
        }
        for (int c = 0; c < C; c += 4) {
            Taps t[NV][4];
            float tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = min(c + u, C - 1);
                tv[u] = tg[cc * HW];
#pragma unroll
                for (int vi = 0; vi < NV; ++vi) t[vi][u] = gather(a.src[vi] + ((int64_t)b * C + cc) * HW, ta[vi]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c + u < C) {
#pragma unroll
                    for (int vi = 0; vi < NV; ++vi) {
                        const float wv = blend(t[vi][u], s[vi]);
                        nz[vi] |= (wv != 0.f);               // loss_functions.py:11  (warped == 0).prod(1)
                        acc[vi] += fabsf(tv[u] - wv);        // :12-13
                    }
                }
            }
        }
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) {
            float m = 1.f;
            if (a.mask) m = fabsf(a.mask[((int64_t)b * NV + vi) * HW + pix]);   // loss_functions_sfm.py:30-31
            lsum[vi] += nz[vi] ? acc[vi] * m : 0.f;
        }
    }
    }
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        const float r = wave_sum(lsum[vi]);
        if (threadIdx.x == 0) red[threadIdx.y][vi] = r;
    }
    __syncthreads();
    if (tid < a.V) {
        const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.partials[blk * a.V + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// Deterministic second stage: one block sums the per-block partials in a fixed order.
__global__ __launch_bounds__(256) void photo_reduce_kernel(const float *partials, int64_t nblk, int V, float inv_n,
                                                           float *loss_out, float *view_loss) {
    __shared__ float red[4];
    float total = 0.f;
    for (int v = 0; v < V; ++v) {
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < nblk; i += 256) s += partials[i * V + v];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float t = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;     // mean over B*C*H*W
        __syncthreads();
        if (threadIdx.x == 0 && view_loss) view_loss[v] = t;
        total += t;
    }
    if (threadIdx.x == 0) loss_out[0] = total;
}

// Reduce 12 values per view over the block and add them to pose_ws[(v*B+b)*12 + k].
// Wave stage = packed butterfly: at every step a lane hands HALF of its values to its partner and keeps the sum of
// the other half, so 24 values cost 12+6+3 exchanges plus 3x3 plain steps (30 cross-lane ops instead of 24 x 6).
template <int NV>
__device__ __forceinline__ void reduce_pose_partials(float (&acc)[NV][12], int V, int B, int b, float *pose_ws,
                                                     float (*red)[DVF_MAX_VIEWS * 12]) {
    constexpr int N = NV * 12;
    float v[N];
#pragma unroll
    for (int vi = 0; vi < NV; ++vi)
#pragma unroll
        for (int k = 0; k < 12; ++k) v[vi * 12 + k] = acc[vi][k];
    const int lane = threadIdx.x;                          // blockDim.x == 64: one wave per threadIdx.y
    // halving steps on lane bits 5, 4, 3 (as long as the count stays even)
    int base = 0;                                          // original index of v[0] in this lane
    constexpr int N1 = N / 2, N2 = (N % 4 == 0) ? N / 4 : N1, N3 = (N % 8 == 0) ? N / 8 : N2;
    {
        const bool up = lane & 32;
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            const float send = up ? v[i] : v[i + N1], keep = up ? v[i + N1] : v[i];
            v[i] = keep + __shfl_xor(send, 32, 64);
        }
        base += up ? N1 : 0;
    }
    if (N2 != N1) {
        const bool up = lane & 16;
#pragma unroll
        for (int i = 0; i < N2; ++i) {
            const float send = up ? v[i] : v[i + N2], keep = up ? v[i + N2] : v[i];
            v[i] = keep + __shfl_xor(send, 16, 64);
        }
        base += up ? N2 : 0;
    } else {
#pragma unroll
        for (int i = 0; i < N1; ++i) v[i] += __shfl_xor(v[i], 16, 64);
    }
    if (N3 != N2) {
        const bool up = lane & 8;
#pragma unroll
        for (int i = 0; i < N3; ++i) {
            const float send = up ? v[i] : v[i + N3], keep = up ? v[i + N3] : v[i];
            v[i] = keep + __shfl_xor(send, 8, 64);
        }
        base += up ? N3 : 0;
    } else {
#pragma unroll
        for (int i = 0; i < N2; ++i) v[i] += __shfl_xor(v[i], 8, 64);
    }
#pragma unroll
    for (int i = 0; i < N3; ++i) {
        float t = v[i];
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 1, 64);
        v[i] = t;
    }
    if ((lane & 7) == 0) {
#pragma unroll
        for (int i = 0; i < N3; ++i) red[threadIdx.y][base + i] = v[i];
    }
    __syncthreads();
    const int tid = threadIdx.y * TX + threadIdx.x;
    if (tid < V * 12) {
        const float r = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        atomicAdd(&pose_ws[((int64_t)(tid / 12) * B + b) * 12 + (tid % 12)], r);
    }
}

template <bool BORDER, bool ALIGN, bool PIX, int NV>
__global__ __launch_bounds__(256) void photo_bwd_kernel(PhotoArgs a) {
    __shared__ ViewGeo geo[DVF_MAX_VIEWS];
    __shared__ float kinv[9], kmat[9];
    __shared__ float red[TY][DVF_MAX_VIEWS * 12];
    const int b = blockIdx.z, tid = threadIdx.y * TX + threadIdx.x;
    block_setup(a, b, tid, geo, kinv, kmat);
    const int x = blockIdx.x * TX + threadIdx.x;
    const int W = a.W, H = a.H, C = a.C;
    const int64_t HW = (int64_t)H * W;
    const float scale = a.grad_loss[0] / ((float)a.B * (float)C * (float)H * (float)W);
    float pacc[NV][12];
#pragma unroll
    for (int vi = 0; vi < NV; ++vi)
#pragma unroll
        for (int k = 0; k < 12; ++k) pacc[vi][k] = 0.f;
    for (int rp = 0; rp < RPT; ++rp) {
    const int y = (blockIdx.y * RPT + rp) * TY + threadIdx.y;
    if (x < W && y < H) {
        const int64_t pix = (int64_t)y * W + x;
        const float d = a.depth[(int64_t)b * HW + pix];
        const float u = (float)x, v = (float)y;
        const float c0x = kinv[0] * u + kinv[1] * v + kinv[2];
        const float c0y = kinv[3] * u + kinv[4] * v + kinv[5];
        const float c0z = kinv[6] * u + kinv[7] * v + kinv[8];
        const float cx = c0x * d, cy = c0y * d, cz = c0z * d;
        const float *tg = a.tgt + (int64_t)b * C * HW + pix;
        float gd = 0.f;
        const bool need_tgt = a.g_tgt != nullptr;
        // pass 1 for ALL views together (see photo_fwd_kernel): d loss / d ix, iy without the validity factor
        Samp sv[NV];
        TapAddr tav[NV];
        float gixv[NV], giyv[NV], absumv[NV], mv[NV];
        bool nzv[NV];
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) {
            sv[vi] = project<BORDER, ALIGN, PIX>(geo[vi], cx, cy, cz, W, H);
            tav[vi] = tap_addr(sv[vi], W, H);
            gixv[vi] = giyv[vi] = absumv[vi] = 0.f;
            nzv[vi] = false;
            mv[vi] = a.mask ? a.mask[((int64_t)b * NV + vi) * HW + pix] : 1.f;
        }
        for (int c = 0; c < C; c += 4) {
            Taps t[NV][4];
            float tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = min(c + u, C - 1);
                tv[u] = tg[cc * HW];
#pragma unroll
                for (int vi = 0; vi < NV; ++vi) t[vi][u] = gather(a.src[vi] + ((int64_t)b * C + cc) * HW, tav[vi]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c + u < C) {
#pragma unroll
                    for (int vi = 0; vi < NV; ++vi) {
                        const float wv = blend(t[vi][u], sv[vi]);
                        nzv[vi] |= (wv != 0.f);
                        const float df = tv[u] - wv;
                        const float sg = sgn(df * mv[vi]);   // sign of the masked difference
                        absumv[vi] += fabsf(df);
                        float dox, doy;
                        blend_grad(t[vi][u], sv[vi], dox, doy);
                        gixv[vi] -= sg * dox;                // d|.|/d warped = -sign
                        giyv[vi] -= sg * doy;
                    }
                }
            }
        }
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) {
            const Samp &s = sv[vi];
            const TapAddr &ta = tav[vi];
            const float *sp = a.src[vi] + (int64_t)b * C * HW;
            const float m = mv[vi], gix = gixv[vi], giy = giyv[vi], absum = absumv[vi];
            const bool nz = nzv[vi];
            const float vm = nz ? m * scale : 0.f;           // validity * explainability * upstream / N
            if (a.g_mask) a.g_mask[((int64_t)b * NV + vi) * HW + pix] = nz ? absum * sgn(m) * scale : 0.f;
            // pass 2 (features only): grad target and scatter-add grad source
            float *gs = a.g_src[vi];
            if (nz && (need_tgt || gs)) {
                for (int c = 0; c < C; ++c) {
                    const Taps t = gather(sp + c * HW, ta);
                    const float df = tg[c * HW] - blend(t, s);
                    const float g = sgn(df * m) * vm;        // d loss / d tgt_c ; d loss / d warped_c = -g
                    if (need_tgt) {
                        float *gt = a.g_tgt + ((int64_t)b * C + c) * HW + pix;
                        *gt = (vi == 0) ? g : (*gt + g);     // same thread owns this element across views
                    }
                    if (gs) {
                        float *gp = gs + ((int64_t)b * C + c) * HW + (int64_t)s.y0 * W + s.x0;
                        const bool xin0 = (unsigned)s.x0 < (unsigned)W, xin1 = (unsigned)(s.x0 + 1) < (unsigned)W;
                        const bool yin0 = (unsigned)s.y0 < (unsigned)H, yin1 = (unsigned)(s.y0 + 1) < (unsigned)H;
                        if (xin0 && yin0) atomicAdd(gp, -g * s.wnw);
                        if (xin1 && yin0) atomicAdd(gp + 1, -g * s.wne);
                        if (xin0 && yin1) atomicAdd(gp + W, -g * s.wsw);
                        if (xin1 && yin1) atomicAdd(gp + W + 1, -g * s.wse);
                    }
                }
            } else if (need_tgt && vi == 0) {
                for (int c = 0; c < C; ++c) a.g_tgt[((int64_t)b * C + c) * HW + pix] = 0.f;
            }
            // chain to the projected point                     cam2pixel, inverse_warp.py:61-66
            const float gxq = gix * vm * s.dix, gyq = giy * vm * s.diy;
            const float gpx = gxq / s.Z, gpy = gyq / s.Z;
            const float gpz = s.zpass ? -(gxq * s.xq + gyq * s.yq) / s.Z : 0.f;
            const ViewGeo &g = geo[vi];
            // d p / d depth = (K R) cam0
            const float gcx = g.A[0] * gpx + g.A[3] * gpy + g.A[6] * gpz;
            const float gcy = g.A[1] * gpx + g.A[4] * gpy + g.A[7] * gpz;
            const float gcz = g.A[2] * gpx + g.A[5] * gpy + g.A[8] * gpz;
            gd += gcx * c0x + gcy * c0y + gcz * c0z;
            // y = R cam + t ; g_y = K^T g_p ; accumulate g_t and g_R = g_y (x) cam
            const float gyx = kmat[0] * gpx + kmat[3] * gpy + kmat[6] * gpz;
            const float gyy = kmat[1] * gpx + kmat[4] * gpy + kmat[7] * gpz;
            const float gyz = kmat[2] * gpx + kmat[5] * gpy + kmat[8] * gpz;
            pacc[vi][0] += gyx; pacc[vi][1] += gyy; pacc[vi][2] += gyz;
            pacc[vi][3] += gyx * cx; pacc[vi][4] += gyx * cy; pacc[vi][5] += gyx * cz;
            pacc[vi][6] += gyy * cx; pacc[vi][7] += gyy * cy; pacc[vi][8] += gyy * cz;
            pacc[vi][9] += gyz * cx; pacc[vi][10] += gyz * cy; pacc[vi][11] += gyz * cz;
        }
        if (a.g_depth) a.g_depth[(int64_t)b * HW + pix] = gd;
    }
    }
    if (a.pose_ws) reduce_pose_partials<NV>(pacc, NV, a.B, b, a.pose_ws, red);
}

// pose_ws[(v*B+b)*12] = (g_t[3], g_R[9]) -> g_pose[(v*B+b)*6] through d R / d (rx,ry,rz).
__global__ void pose_finalize_kernel(const float *pose, const float *ws, float *g_pose, int n, uint32_t quat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pose + (int64_t)i * 6;
    const float *g = ws + (int64_t)i * 12;
    const float *gR = g + 3;
    float *o = g_pose + (int64_t)i * 6;
    if (quat == 2) {
        // se3 (w, u): y = R x + R u.  g_u = R^T g_t;  dL/dR += g_t (x) u;  dL/dw_i = <dL/dR, dR/dw_i> with
        // dR/dw_i = (w_i [w]x + [w x (I - R) e_i]x) / th^2 * R   (se3_generate.py:57-100)
        float R[9];
        pose_to_R(p, 2, R);
        const float gt[3] = {g[0], g[1], g[2]};
        const float u[3] = {p[3], p[4], p[5]}, w[3] = {p[0], p[1], p[2]};
        float GR[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) GR[i * 3 + j] = gR[i * 3 + j] + gt[i] * u[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) o[3 + j] = R[0 * 3 + j] * gt[0] + R[1 * 3 + j] * gt[1] + R[2 * 3 + j] * gt[2];
        const float th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
        const float Wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        for (int i = 0; i < 3; ++i) {
            float D[9];
            if (th2 < 1e-12f) {
                const float G[3][9] = {{0, 0, 0, 0, 0, 1, 0, -1, 0}, {0, 0, -1, 0, 0, 0, 1, 0, 0}, {0, 1, 0, -1, 0, 0, 0, 0, 0}};
                for (int e = 0; e < 9; ++e) D[e] = G[i][e];       // generators exactly as the reference writes them
            } else {
                // v = (I - R) e_i ; c = w x v
                const float v[3] = {(i == 0 ? 1.f : 0.f) - R[0 * 3 + i], (i == 1 ? 1.f : 0.f) - R[1 * 3 + i],
                                    (i == 2 ? 1.f : 0.f) - R[2 * 3 + i]};
                const float cx = w[1] * v[2] - w[2] * v[1], cy = w[2] * v[0] - w[0] * v[2], cz = w[0] * v[1] - w[1] * v[0];
                const float Cx[9] = {0, -cz, cy, cz, 0, -cx, -cy, cx, 0};
                float M[9];
                for (int e = 0; e < 9; ++e) M[e] = (w[i] * Wx[e] + Cx[e]) / th2;
                mat3mul(M, R, D);
            }
            float sacc = 0.f;
            for (int e = 0; e < 9; ++e) sacc += GR[e] * D[e];
            o[i] = sacc;
        }
        return;
    }
    o[0] = g[0]; o[1] = g[1]; o[2] = g[2];
    if (!quat) {
        const float cx = cosf(p[3]), sx = sinf(p[3]);
        const float cy = cosf(p[4]), sy = sinf(p[4]);
        const float cz = cosf(p[5]), sz = sinf(p[5]);
        const float X[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
        const float Y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
        const float Z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
        const float dX[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx};
        const float dY[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy};
        const float dZ[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
        float T[9], D[9];
        const float *M[3][3] = {{dX, Y, Z}, {X, dY, Z}, {X, Y, dZ}};
        for (int k = 0; k < 3; ++k) {
            mat3mul(M[k][0], M[k][1], T);
            mat3mul(T, M[k][2], D);
            float s = 0.f;
            for (int e = 0; e < 9; ++e) s += gR[e] * D[e];
            o[3 + k] = s;
        }
    } else {
        const float n2 = 1.f + p[3] * p[3] + p[4] * p[4] + p[5] * p[5];
        const float nn = sqrtf(n2);
        const float q[4] = {1.f / nn, p[3] / nn, p[4] / nn, p[5] / nn};
        const float w = q[0], x = q[1], y = q[2], z = q[3];
        const float dW[9] = {2 * w, -2 * z, 2 * y, 2 * z, 2 * w, -2 * x, -2 * y, 2 * x, 2 * w};
        const float dXq[9] = {2 * x, 2 * y, 2 * z, 2 * y, -2 * x, -2 * w, 2 * z, 2 * w, -2 * x};
        const float dYq[9] = {-2 * y, 2 * x, 2 * w, 2 * x, 2 * y, 2 * z, -2 * w, 2 * z, -2 * y};
        const float dZq[9] = {-2 * z, -2 * w, 2 * x, 2 * w, -2 * z, 2 * y, 2 * x, 2 * y, 2 * z};
        const float *D[4] = {dW, dXq, dYq, dZq};
        float gq[4], dot = 0.f;
        for (int k = 0; k < 4; ++k) {
            float s = 0.f;
            for (int e = 0; e < 9; ++e) s += gR[e] * D[k][e];
            gq[k] = s;
            dot += s * q[k];
        }
        for (int k = 0; k < 3; ++k) o[3 + k] = (gq[k + 1] - q[k + 1] * dot) / nn;   // through q = u / |u|
    }
}

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

template <typename F>
int dispatch_mode(uint32_t flags, F &&f) {
    using T = std::integral_constant<bool, true>;
    using N = std::integral_constant<bool, false>;
    if (flags & DVF_PIXEL_COORDS) return f(N{}, N{}, T{});
    const bool border = flags & DVF_PAD_BORDER, align = flags & DVF_ALIGN_CORNERS;
    if (border && align) return f(T{}, T{}, N{});
    if (border) return f(T{}, N{}, N{});
    if (align) return f(N{}, T{}, N{});
    return f(N{}, N{}, N{});
}

inline uint32_t rot_mode(uint32_t flags) { return (flags & DVF_POSE_SE3) ? 2u : ((flags & DVF_ROT_QUAT) ? 1u : 0u); }

inline dim3 pix_grid(int B, int H, int W) { return dim3((W + TX - 1) / TX, (H + TY - 1) / TY, B); }
inline dim3 photo_grid(int B, int H, int W) { return dim3((W + TX - 1) / TX, (H + TY * RPT - 1) / (TY * RPT), B); }

}  // namespace

extern "C" {

int64_t dvf_pose_ws_floats(int V, int B) { return (int64_t)V * B * 12; }

int64_t dvf_photo_partials_floats(int B, int H, int W, int V) {
    const dim3 g = photo_grid(B, H, W);
    return (int64_t)g.x * g.y * g.z * V;
}

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

static int fill_photo_args(PhotoArgs &a, const float *tgt, const float *const *srcs, int V, const float *depth,
                           const float *pose, const float *K, const float *Kinv, const float *mask, int B, int C,
                           int H, int W, uint32_t flags) {
    if (!tgt || !srcs || !depth || !pose || !K || !Kinv || V < 1 || V > DVF_MAX_VIEWS || B <= 0 || C <= 0 ||
        H < 2 || W < 2 || B > 65535)
        return DVF_ERR_INVALID_ARG;
    a = PhotoArgs{};
    a.tgt = tgt;
    for (int v = 0; v < V; ++v) {
        if (!srcs[v]) return DVF_ERR_INVALID_ARG;
        a.src[v] = srcs[v];
    }
    a.depth = depth; a.pose = pose; a.K = K; a.Kinv = Kinv; a.mask = mask;
    a.B = B; a.C = C; a.H = H; a.W = W; a.V = V;
    a.quat = rot_mode(flags);
    return DVF_OK;
}

int dvf_photo_loss_fwd(const float *tgt, const float *const *srcs, int V, const float *depth, const float *pose,
                       const float *K, const float *Kinv, const float *mask, float *loss_out, float *view_loss,
                       float *partials, int B, int C, int H, int W, uint32_t flags, void *stream) {
    PhotoArgs a;
    int rc = fill_photo_args(a, tgt, srcs, V, depth, pose, K, Kinv, mask, B, C, H, W, flags);
    if (rc != DVF_OK) return rc;
    if (!loss_out || !partials) return DVF_ERR_INVALID_ARG;
    a.partials = partials;
    hipStream_t st = dvf_stream(stream);
    const dim3 grid = photo_grid(B, H, W);
    rc = dispatch_mode(flags, [&](auto border, auto align, auto pix) {
        constexpr bool BD = decltype(border)::value, AL = decltype(align)::value, PX = decltype(pix)::value;
        switch (V) {
            case 1: photo_fwd_kernel<BD, AL, PX, 1><<<grid, dim3(TX, TY), 0, st>>>(a); break;
            case 2: photo_fwd_kernel<BD, AL, PX, 2><<<grid, dim3(TX, TY), 0, st>>>(a); break;
            case 3: photo_fwd_kernel<BD, AL, PX, 3><<<grid, dim3(TX, TY), 0, st>>>(a); break;
            default: photo_fwd_kernel<BD, AL, PX, 4><<<grid, dim3(TX, TY), 0, st>>>(a); break;
        }
        DVF_LAUNCH_CHECK();
        return DVF_OK;
    });
    if (rc != DVF_OK) return rc;
    const float inv_n = 1.f / ((float)B * (float)C * (float)H * (float)W);
    photo_reduce_kernel<<<1, 256, 0, st>>>(partials, (int64_t)grid.x * grid.y * grid.z, V, inv_n, loss_out, view_loss);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_photo_loss_bwd(const float *tgt, const float *const *srcs, int V, const float *depth, const float *pose,
                       const float *K, const float *Kinv, const float *mask, const float *grad_loss, float *g_depth,
                       float *g_pose, float *g_tgt, float *const *g_srcs, float *g_mask, float *pose_ws, int B,
                       int C, int H, int W, uint32_t flags, void *stream) {
    PhotoArgs a;
    int rc = fill_photo_args(a, tgt, srcs, V, depth, pose, K, Kinv, mask, B, C, H, W, flags);
    if (rc != DVF_OK) return rc;
    if (!grad_loss || (g_pose && !pose_ws) || (g_mask && !mask)) return DVF_ERR_INVALID_ARG;
    a.grad_loss = grad_loss;
    a.g_depth = g_depth; a.g_tgt = g_tgt; a.g_mask = g_mask;
    a.pose_ws = g_pose ? pose_ws : nullptr;
    for (int v = 0; v < V; ++v) a.g_src[v] = g_srcs ? g_srcs[v] : nullptr;
    hipStream_t st = dvf_stream(stream);
    if (g_pose && hipMemsetAsync(pose_ws, 0, sizeof(float) * 12 * V * B, st) != hipSuccess) return DVF_ERR_LAUNCH;
    rc = dispatch_mode(flags, [&](auto border, auto align, auto pix) {
        constexpr bool BD = decltype(border)::value, AL = decltype(align)::value, PX = decltype(pix)::value;
        const dim3 grid = photo_grid(B, H, W);
        switch (V) {
            case 1: photo_bwd_kernel<BD, AL, PX, 1><<<grid, dim3(TX, TY), 0, st>>>(a); break;
            case 2: photo_bwd_kernel<BD, AL, PX, 2><<<grid, dim3(TX, TY), 0, st>>>(a); break;
            case 3: photo_bwd_kernel<BD, AL, PX, 3><<<grid, dim3(TX, TY), 0, st>>>(a); break;
            default: photo_bwd_kernel<BD, AL, PX, 4><<<grid, dim3(TX, TY), 0, st>>>(a); break;
        }
        DVF_LAUNCH_CHECK();
        return DVF_OK;
    });
    if (rc != DVF_OK) return rc;
    if (g_pose) {
        pose_finalize_kernel<<<(V * B + 63) / 64, 64, 0, st>>>(pose, pose_ws, g_pose, V * B, rot_mode(flags));
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
