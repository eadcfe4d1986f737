// Device helpers shared by the warp / photometric-loss translation units: pose -> rotation, the pixel -> cam -> SE3 ->
// pixel chain of one (pixel, view), 2x2 tap addressing, bilinear blend and its gradient, the pose-gradient finalisation.
// Arithmetic follows the reference op by op (file:line in each helper) in fp32; products/sums whose rounding decides an
// exact comparison in the reference are written with __f*_rn so hipcc does not contract them into FMAs.
#pragma once
#include "dvf_common.h"

namespace dvfw {


constexpr int TX = 64;   // pixels along a row per wave
constexpr int TY = 4;    // rows per block pass (one wave each)

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
static __device__ void pose_to_R(const float *p, int mode, float *R) {
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

static __device__ void build_view(const float *pose6, const float *K, int mode, ViewGeo *g) {
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

// Reduce 12 values per view over the block and add them to pose_ws[(v*B+b)*12 + k].
// Wave stage = packed butterfly: at every step a lane hands HALF of its values to its partner and keeps the sum of
// the other half, so 24 values cost 12+6+3 exchanges plus 3x3 plain steps (30 cross-lane ops instead of 24 x 6).
// `part` != nullptr: the block's sums are STORED to part[0 .. V*12) (deterministic two-stage reduction) instead.
template <int NV>
__device__ __forceinline__ void reduce_pose_partials(float (&acc)[NV][12], int V, int B, int b, float *pose_ws,
                                                     float (*red)[DVF_MAX_VIEWS * 12], float *part = nullptr) {
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
        if (part) part[tid] = r;
        else atomicAdd(&pose_ws[((int64_t)(tid / 12) * B + b) * 12 + (tid % 12)], r);
    }
}


// pose_ws[(v*B+b)*12] = (g_t[3], g_R[9]) -> g_pose[(v*B+b)*6] through d R / d (rx,ry,rz).
static __global__ void pose_finalize_kernel(const float *pose, const float *ws, float *g_pose, int n, uint32_t quat) {
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

// ---- host side: template dispatch on (padding, align_corners, pixel-coordinate) and the rotation mode of the flags
template <typename F>
inline int dispatch_mode(uint32_t flags, F &&f) {
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

}  // namespace dvfw
