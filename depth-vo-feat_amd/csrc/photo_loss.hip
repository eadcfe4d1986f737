// Fused inverse-warp + photometric-L1 loss (forward and backward), one launch per pyramid scale for all V views.
//
// A block owns a 64 x 4 tile of TARGET pixels (one pixel per lane, one image row per wave).  Per pixel the
// pixel -> cam -> SE3 -> pixel chain of every view is evaluated in registers (depth and target are read once).  The
// 2x2 source neighbourhoods of a tile's pixels form a slightly larger, displaced tile of the source image: its bounding
// box (the "footprint") is reduced over the block, and
//   * forward / backward pass 1: the footprint of PT_CC channels is STAGED IN LDS with coalesced 16-byte row loads and
//     every pixel reads its four taps from LDS (no per-pixel gathers from HBM);
//   * backward pass 2 (source gradients, the feature-reconstruction loss): the scatter-add of the four tap weights is
//     ACCUMULATED IN THE SAME LDS TILE (ds_add_f32) and flushed once per footprint element with row-contiguous global
//     atomics -- four global atomics per (pixel, channel, view) become ~1.3, in 256-byte contiguous shapes.
// A footprint that does not fit the LDS tile (wild depth inside one tile) falls back, for that (block, view) only, to
// direct 8-byte pair gathers and direct global atomics; a footprint with no valid tap at all is skipped.
// The loss sum and the pose-gradient partials are reduced wavefront-first and finished in fixed order by a second
// stage (deterministic: no float atomics on either).
//
// Arithmetic follows the reference op by op (file:line in warp_common.h) in fp32.  `in_scale` multiplies every image
// value as it is loaded (fl(in_scale * x), exactly the reference's `0.004 * img` at unsupervise.py:101).
#include "warp_common.h"

namespace {

using namespace dvfw;

constexpr int PT_CC = 8;          // channels per staged chunk
constexpr int PT_CAP = 1200;      // floats per channel of a footprint tile (68 x 5 typical; up to 128 x 9 / 72 x 16): 8 channels = 38.4 KB, four blocks per CU
constexpr int PT_MAXC = 32;       // the backward kernel keeps one L1 sign per (view, channel) in two 32-bit fields

struct PhotoArgs {
    const float *tgt;
    const float *src[DVF_MAX_VIEWS];
    const float *depth, *pose, *K, *Kinv, *mask;
    float *partials;
    // backward
    const float *grad_loss;
    float *g_depth, *g_tgt, *g_mask, *pose_part;
    float *g_src[DVF_MAX_VIEWS];
    int B, C, H, W, V;
    uint32_t quat;
    float in_scale;
    int vec;                       // sources may be staged in 16-byte lanes (W % 4 == 0, 16-byte aligned bases)
    int dbg;                       // ablation switches (-DDVF_TUNING builds only): 1 no pass 2, 2 no flush atomics, 4 no LDS adds, 8 no staging
};

__device__ __forceinline__ void block_setup(const PhotoArgs &a, int b, int tid, ViewGeo *geo, float *kinv, float *kmat) {
    if (tid < a.V) build_view(a.pose + ((int64_t)tid * a.B + b) * 6, a.K + (int64_t)b * 9, (int)a.quat, &geo[tid]);
    if (tid >= 64 && tid < 73) kinv[tid - 64] = a.Kinv[(int64_t)b * 9 + tid - 64];
    if (kmat && tid >= 128 && tid < 137) kmat[tid - 128] = a.K[(int64_t)b * 9 + tid - 128];
    __syncthreads();
}

// Clamped 2x2 tap position of one (pixel, view): the two taps of a row are the adjacent pair (xb, xb+1), always inside
// the row; `straight` tells whether the pair starts at x0 or one to its right/left (x0 = -1 or W-1).
struct TapPos { int xb, y0, y1; bool straight, v_nw, v_ne, v_sw, v_se, any; };

__device__ __forceinline__ TapPos tap_pos(const Samp &s, int W, int H, bool inside) {
    TapPos p;
    const bool xin0 = (unsigned)s.x0 < (unsigned)W, xin1 = (unsigned)(s.x0 + 1) < (unsigned)W;
    const bool yin0 = (unsigned)s.y0 < (unsigned)H, yin1 = (unsigned)(s.y0 + 1) < (unsigned)H;
    p.xb = min(max(s.x0, 0), W - 2);
    p.y0 = min(max(s.y0, 0), H - 1);
    p.y1 = min(max(s.y0 + 1, 0), H - 1);
    p.straight = (s.x0 == p.xb);
    p.v_nw = inside && xin0 && yin0; p.v_ne = inside && xin1 && yin0;
    p.v_sw = inside && xin0 && yin1; p.v_se = inside && xin1 && yin1;
    p.any = p.v_nw || p.v_ne || p.v_sw || p.v_se;
    return p;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// Footprint of the block's pixels in one source view (block-uniform).
struct Foot { int xlo, ylo, RS, rows, mode; };       // mode 0: no valid tap, 1: LDS tile, 2: direct (does not fit)

template <int NV>
__device__ __forceinline__ void footprints(const TapPos (&tp)[NV], int W, int vec, int (*bbs)[DVF_MAX_VIEWS][4],
                                           Foot (&ft)[NV]) {
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        const TapPos &p = tp[vi];
        const int mnx = wave_min_i(p.any ? p.xb : 0x7fffffff), mxx = wave_max_i(p.any ? p.xb + 1 : -1);
        const int mny = wave_min_i(p.any ? p.y0 : 0x7fffffff), mxy = wave_max_i(p.any ? p.y1 : -1);
        if (threadIdx.x == 0) {
            bbs[threadIdx.y][vi][0] = mnx; bbs[threadIdx.y][vi][1] = mxx;
            bbs[threadIdx.y][vi][2] = mny; bbs[threadIdx.y][vi][3] = mxy;
        }
    }
    __syncthreads();
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        int mnx = bbs[0][vi][0], mxx = bbs[0][vi][1], mny = bbs[0][vi][2], mxy = bbs[0][vi][3];
#pragma unroll
        for (int w = 1; w < TY; ++w) {
            mnx = min(mnx, bbs[w][vi][0]); mxx = max(mxx, bbs[w][vi][1]);
            mny = min(mny, bbs[w][vi][2]); mxy = max(mxy, bbs[w][vi][3]);
        }
        Foot f;
        // (readfirstlane: the values are block-uniform; keeps the staging loops' bookkeeping scalar)
        mnx = __builtin_amdgcn_readfirstlane(mnx); mxx = __builtin_amdgcn_readfirstlane(mxx);
        mny = __builtin_amdgcn_readfirstlane(mny); mxy = __builtin_amdgcn_readfirstlane(mxy);
        f.xlo = vec ? (mnx & ~3) : mnx;
        f.ylo = mny;
        f.RS = vec ? ((mxx - f.xlo + 1 + 3) & ~3) : (mxx - f.xlo + 1);
        f.rows = mxy - mny + 1;
        f.mode = (mxx < 0) ? 0 : ((f.RS * f.rows <= PT_CAP && f.RS <= 256) ? 1 : 2);
        ft[vi] = f;
    }
}

// idx / d for 0 <= idx < 2^16, small d, through fp32 (exact: the +0.5 keeps the quotient 0.5/d away from an integer)
__device__ __forceinline__ int div_small(int idx, float inv_d) { return (int)(((float)idx + 0.5f) * inv_d); }

// Stage the footprint of channels c0 .. c0+nch-1 of image b of `src` into tile[ch][row][RS], values scaled by in_scale.
__device__ __forceinline__ void stage_tile(float *tile, const float *__restrict__ src, const Foot &f, int b, int C, int c0,
                                           int nch, int H, int W, float in_scale, int vec, int tid) {
    const int chs = f.rows * f.RS;
    const float *base = src + ((int64_t)b * C + c0) * H * W + (int64_t)f.ylo * W + f.xlo;
    if (vec) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int QW = f.RS >> 2, nrow = nch * f.rows, total = nrow * QW;
        const float inv_qw = 1.0f / (float)QW, inv_rows = 1.0f / (float)f.rows;
        constexpr int U = 4;                                 // independent loads in flight per thread
        for (int i0 = tid; i0 < total; i0 += 256 * U) {
            f4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + 256 * u;
                dst[u] = -1;
                if (idx < total) {
                    const int r = div_small(idx, inv_qw), q = idx - r * QW;
                    const int ch = div_small(r, inv_rows), y = r - ch * f.rows;
                    dst[u] = ch * chs + y * f.RS + 4 * q;
                    if (f.xlo + 4 * q < W) v[u] = *reinterpret_cast<const f4 *>(base + ((int64_t)ch * H + y) * W + 4 * q);
                    else v[u] = f4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) {
                    f4 t = v[u];
                    if (in_scale != 1.f) { t.x = __fmul_rn(in_scale, t.x); t.y = __fmul_rn(in_scale, t.y); t.z = __fmul_rn(in_scale, t.z); t.w = __fmul_rn(in_scale, t.w); }
                    *reinterpret_cast<f4 *>(tile + dst[u]) = t;
                }
        }
    } else {
        const int total = nch * chs;
        const float inv_rs = 1.0f / (float)f.RS, inv_rows = 1.0f / (float)f.rows;
        for (int idx = tid; idx < total; idx += 256) {
            const int r = div_small(idx, inv_rs), x = idx - r * f.RS;
            const int ch = div_small(r, inv_rows), y = r - ch * f.rows;
            float v = (f.xlo + x < W) ? base[((int64_t)ch * H + y) * W + x] : 0.f;
            if (in_scale != 1.f) v = __fmul_rn(in_scale, v);
            tile[idx] = v;
        }
    }
}

// Tile-relative offsets of one pixel's two tap pairs (0 when the pixel has no valid tap: reads stay inside the tile).
struct TileOff { int top, bot; };
__device__ __forceinline__ TileOff tile_off(const TapPos &p, const Foot &f) {
    TileOff o;
    o.top = p.any ? (p.y0 - f.ylo) * f.RS + (p.xb - f.xlo) : 0;
    o.bot = p.any ? (p.y1 - f.ylo) * f.RS + (p.xb - f.xlo) : 0;
    return o;
}

__device__ __forceinline__ Taps taps_select(float t0, float t1, float b0, float b1, const TapPos &p) {
    Taps t;
    t.nw = p.v_nw ? (p.straight ? t0 : t1) : 0.f;
    t.ne = p.v_ne ? (p.straight ? t1 : t0) : 0.f;
    t.sw = p.v_sw ? (p.straight ? b0 : b1) : 0.f;
    t.se = p.v_se ? (p.straight ? b1 : b0) : 0.f;
    return t;
}

__device__ __forceinline__ Taps taps_lds(const float *tch, const TileOff &o, const TapPos &p) {
    return taps_select(tch[o.top], tch[o.top + 1], tch[o.bot], tch[o.bot + 1], p);
}

__device__ __forceinline__ Taps taps_direct(const float *__restrict__ plane, const TapPos &p, int W, float in_scale) {
    const Pair top = *reinterpret_cast<const Pair *>(plane + p.y0 * W + p.xb);
    const Pair bot = *reinterpret_cast<const Pair *>(plane + p.y1 * W + p.xb);
    if (in_scale != 1.f)
        return taps_select(__fmul_rn(in_scale, top.x), __fmul_rn(in_scale, top.y), __fmul_rn(in_scale, bot.x),
                           __fmul_rn(in_scale, bot.y), p);
    return taps_select(top.x, top.y, bot.x, bot.y, p);
}

template <bool BORDER, bool ALIGN, bool PIX, int NV>
__global__ __launch_bounds__(256) void photo_fwd_kernel(PhotoArgs a) {
    __shared__ ViewGeo geo[DVF_MAX_VIEWS];
    __shared__ float kinv[9];
    __shared__ float red[TY][DVF_MAX_VIEWS];
    __shared__ int bbs[TY][DVF_MAX_VIEWS][4];
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.z, tid = threadIdx.y * TX + threadIdx.x;
    block_setup(a, b, tid, geo, kinv, nullptr);
    const int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
    const int W = a.W, H = a.H, C = a.C;
    const int64_t HW = (int64_t)H * W;
    const bool inside = x < W && y < H;
    const int64_t pix = (int64_t)min(y, H - 1) * W + min(x, W - 1);
    const float d = a.depth[(int64_t)b * HW + pix];
    // cam = (Kinv @ (u, v, 1)) * depth                      inverse_warp.py:38-40
    const float u = (float)x, v = (float)y;
    const float cx = (kinv[0] * u + kinv[1] * v + kinv[2]) * d;
    const float cy = (kinv[3] * u + kinv[4] * v + kinv[5]) * d;
    const float cz = (kinv[6] * u + kinv[7] * v + kinv[8]) * d;
    Samp s[NV];
    TapPos tp[NV];
    Foot ft[NV];
    float acc[NV];
    bool nz[NV];
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        s[vi] = project<BORDER, ALIGN, PIX>(geo[vi], cx, cy, cz, W, H);
        tp[vi] = tap_pos(s[vi], W, H, inside);
        acc[vi] = 0.f;
        nz[vi] = false;
    }
    footprints<NV>(tp, W, a.vec, bbs, ft);
    const float *tg = a.tgt + (int64_t)b * C * HW + pix;
    for (int c0 = 0; c0 < C; c0 += PT_CC) {
        const int nch = min(PT_CC, C - c0);
        float tv[PT_CC];
#pragma unroll
        for (int k = 0; k < PT_CC; ++k) {
            tv[k] = (k < nch) ? tg[(int64_t)(c0 + k) * HW] : 0.f;
            if (a.in_scale != 1.f) tv[k] = __fmul_rn(a.in_scale, tv[k]);
        }
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) {
            const Foot f = ft[vi];
            if (f.mode == 0) continue;                      // no pixel of the block samples inside this source
            if (f.mode == 1) {
                __syncthreads();                            // previous tile fully consumed
                if (!DVF_DBG(a, 8)) stage_tile(tile, a.src[vi], f, b, C, c0, nch, H, W, a.in_scale, a.vec, tid);
                __syncthreads();
            }
            if (!tp[vi].any) continue;                      // warped == 0 in every channel: nz stays false
            const TileOff o = tile_off(tp[vi], f);
            const int chs = f.rows * f.RS;
            const float *sp = a.src[vi] + ((int64_t)b * C + c0) * HW;
#pragma unroll
            for (int k = 0; k < PT_CC; ++k) {
                if (k < nch) {
                    const Taps t = (f.mode == 1) ? taps_lds(tile + k * chs, o, tp[vi]) : taps_direct(sp + (int64_t)k * HW, tp[vi], W, a.in_scale);
                    const float wv = blend(t, s[vi]);
                    nz[vi] |= (wv != 0.f);                  // loss_functions.py:11  (warped == 0).prod(1)
                    acc[vi] += fabsf(tv[k] - wv);           // :12-13
                }
            }
        }
    }
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        float m = 1.f;
        if (a.mask) m = fabsf(a.mask[((int64_t)b * NV + vi) * HW + pix]);      // loss_functions_sfm.py:30-31
        const float r = wave_sum((inside && nz[vi]) ? acc[vi] * m : 0.f);
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

// L1 sign of channel c of a view, two bits per channel in (lo: c < 16, hi: c >= 16): 0 -> 0, 1 -> +1, 2 -> -1.
__device__ __forceinline__ void sign_put(uint32_t &lo, uint32_t &hi, int c, float sg) {
    const uint32_t code = (sg > 0.f) ? 1u : ((sg < 0.f) ? 2u : 0u);
    if (c < 16) lo |= code << (2 * c); else hi |= code << (2 * (c - 16));
}
__device__ __forceinline__ float sign_get(uint32_t lo, uint32_t hi, int c) {
    const uint32_t code = ((c < 16) ? (lo >> (2 * c)) : (hi >> (2 * (c - 16)))) & 3u;
    return (code == 1u) ? 1.f : ((code == 2u) ? -1.f : 0.f);
}

template <bool BORDER, bool ALIGN, bool PIX, int NV>
__global__ __launch_bounds__(256) void photo_bwd_kernel(PhotoArgs a) {
    __shared__ ViewGeo geo[DVF_MAX_VIEWS];
    __shared__ float kinv[9], kmat[9];
    __shared__ float red[TY][DVF_MAX_VIEWS * 12];
    __shared__ int bbs[TY][DVF_MAX_VIEWS][4];
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.z, tid = threadIdx.y * TX + threadIdx.x;
    block_setup(a, b, tid, geo, kinv, kmat);
    const int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
    const int W = a.W, H = a.H, C = a.C;
    const int64_t HW = (int64_t)H * W;
    const bool inside = x < W && y < H;
    const int64_t pix = (int64_t)min(y, H - 1) * W + min(x, W - 1);
    const float scale = a.grad_loss[0] / ((float)a.B * (float)C * (float)H * (float)W);
    const float d = a.depth[(int64_t)b * HW + pix];
    const float u = (float)x, v = (float)y;
    const float c0x = kinv[0] * u + kinv[1] * v + kinv[2];
    const float c0y = kinv[3] * u + kinv[4] * v + kinv[5];
    const float c0z = kinv[6] * u + kinv[7] * v + kinv[8];
    const float cx = c0x * d, cy = c0y * d, cz = c0z * d;
    const float *tg = a.tgt + (int64_t)b * C * HW + pix;
    const bool need_tgt = a.g_tgt != nullptr;
    bool need_src = false;
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) need_src |= a.g_src[vi] != nullptr;
    Samp sv[NV];
    TapPos tp[NV];
    Foot ft[NV];
    float gixv[NV], giyv[NV], absumv[NV], mv[NV];
    uint32_t slo[NV], shi[NV];
    bool nzv[NV];
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        sv[vi] = project<BORDER, ALIGN, PIX>(geo[vi], cx, cy, cz, W, H);
        tp[vi] = tap_pos(sv[vi], W, H, inside);
        gixv[vi] = giyv[vi] = absumv[vi] = 0.f;
        slo[vi] = shi[vi] = 0u;
        nzv[vi] = false;
        mv[vi] = a.mask ? a.mask[((int64_t)b * NV + vi) * HW + pix] : 1.f;
    }
    footprints<NV>(tp, W, a.vec, bbs, ft);
    // ---- pass 1: d loss / d ix, iy (without the validity factor), |diff| sums, exact-zero flags and the L1 signs
    for (int c0 = 0; c0 < C; c0 += PT_CC) {
        const int nch = min(PT_CC, C - c0);
        float tv[PT_CC];
#pragma unroll
        for (int k = 0; k < PT_CC; ++k) {
            tv[k] = (k < nch) ? tg[(int64_t)(c0 + k) * HW] : 0.f;
            if (a.in_scale != 1.f) tv[k] = __fmul_rn(a.in_scale, tv[k]);
        }
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) {
            const Foot f = ft[vi];
            if (f.mode == 0) continue;
            if (f.mode == 1) {
                __syncthreads();
                if (!DVF_DBG(a, 8)) stage_tile(tile, a.src[vi], f, b, C, c0, nch, H, W, a.in_scale, a.vec, tid);
                __syncthreads();
            }
            if (!tp[vi].any) continue;
            const TileOff o = tile_off(tp[vi], f);
            const int chs = f.rows * f.RS;
            const float *sp = a.src[vi] + ((int64_t)b * C + c0) * HW;
#pragma unroll
            for (int k = 0; k < PT_CC; ++k) {
                if (k < nch) {
                    const Taps t = (f.mode == 1) ? taps_lds(tile + k * chs, o, tp[vi]) : taps_direct(sp + (int64_t)k * HW, tp[vi], W, a.in_scale);
                    const float wv = blend(t, sv[vi]);
                    nzv[vi] |= (wv != 0.f);
                    const float df = tv[k] - wv;
                    const float sg = sgn(df * mv[vi]);      // sign of the masked difference
                    absumv[vi] += fabsf(df);
                    float dox, doy;
                    blend_grad(t, sv[vi], dox, doy);
                    gixv[vi] -= sg * dox;                   // d|.|/d warped = -sign
                    giyv[vi] -= sg * doy;
                    sign_put(slo[vi], shi[vi], c0 + k, sg);
                }
            }
        }
    }
    // ---- per view: explainability-mask gradient, chain to depth and to the [R|t] partials
    float gd = 0.f;
    float vmv[NV];
    float pacc[NV][12];
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        const Samp &s = sv[vi];
        const float m = mv[vi];
        const bool nz = inside && nzv[vi];
        const float vm = nz ? m * scale : 0.f;              // validity * explainability * upstream / N
        vmv[vi] = vm;
        if (a.g_mask && inside) a.g_mask[((int64_t)b * NV + vi) * HW + pix] = nz ? absumv[vi] * sgn(m) * scale : 0.f;
        // chain to the projected point                       cam2pixel, inverse_warp.py:61-66
        const float gxq = gixv[vi] * vm * s.dix, gyq = giyv[vi] * vm * s.diy;
        const float gpx = gxq / s.Z, gpy = gyq / s.Z;
        const float gpz = s.zpass ? -(gxq * s.xq + gyq * s.yq) / s.Z : 0.f;
        const ViewGeo &g = geo[vi];
        // d p / d depth = (K R) cam0
        const float gcx = g.A[0] * gpx + g.A[3] * gpy + g.A[6] * gpz;
        const float gcy = g.A[1] * gpx + g.A[4] * gpy + g.A[7] * gpz;
        const float gcz = g.A[2] * gpx + g.A[5] * gpy + g.A[8] * gpz;
        gd += nz ? gcx * c0x + gcy * c0y + gcz * c0z : 0.f;
        // y = R cam + t ; g_y = K^T g_p ; g_t and g_R = g_y (x) cam
        const float gyx = nz ? kmat[0] * gpx + kmat[3] * gpy + kmat[6] * gpz : 0.f;
        const float gyy = nz ? kmat[1] * gpx + kmat[4] * gpy + kmat[7] * gpz : 0.f;
        const float gyz = nz ? kmat[2] * gpx + kmat[5] * gpy + kmat[8] * gpz : 0.f;
        pacc[vi][0] = gyx; pacc[vi][1] = gyy; pacc[vi][2] = gyz;
        pacc[vi][3] = gyx * cx; pacc[vi][4] = gyx * cy; pacc[vi][5] = gyx * cz;
        pacc[vi][6] = gyy * cx; pacc[vi][7] = gyy * cy; pacc[vi][8] = gyy * cz;
        pacc[vi][9] = gyz * cx; pacc[vi][10] = gyz * cy; pacc[vi][11] = gyz * cz;
    }
    if (a.g_depth && inside) a.g_depth[(int64_t)b * HW + pix] = gd;
    if (a.pose_part) {
        const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        reduce_pose_partials<NV>(pacc, NV, a.B, b, nullptr, red, a.pose_part + blk * (NV * 12));
    }
    // ---- grad target: d loss / d tgt_c = sum over views of sign * validity (x in_scale: the kernel scaled the input)
    if (need_tgt && inside) {
        float *gt = a.g_tgt + (int64_t)b * C * HW + pix;
        for (int c = 0; c < C; ++c) {
            float g = 0.f;
#pragma unroll
            for (int vi = 0; vi < NV; ++vi) g += sign_get(slo[vi], shi[vi], c) * vmv[vi];
            gt[(int64_t)c * HW] = g * a.in_scale;
        }
    }
    if (!need_src || DVF_DBG(a, 1)) return;                 // (block-uniform)
    // ---- pass 2: grad source = scatter-add of -g * (bilinear weights), accumulated in the LDS tile per footprint
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int tot4 = (min(PT_CC, C) * PT_CAP) >> 2;
        __syncthreads();                                    // (the last staged tile is still being read by slower waves)
        for (int i = tid; i < tot4; i += 256) reinterpret_cast<f4 *>(tile)[i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c0 = 0; c0 < C; c0 += PT_CC) {
        const int nch = min(PT_CC, C - c0);
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) {
            const Foot f = ft[vi];
            float *gs = a.g_src[vi];
            if (!gs || f.mode == 0) continue;               // (block-uniform)
            const Samp &s = sv[vi];
            const TapPos &p = tp[vi];
            const bool act = p.any && vmv[vi] != 0.f;
            const float gsc = -vmv[vi] * a.in_scale;
            if (f.mode == 2) {                              // footprint too large for the tile: direct global atomics
                if (act) {
                    float *gp = gs + ((int64_t)b * C + c0) * HW + (int64_t)s.y0 * W + s.x0;
                    for (int k = 0; k < nch; ++k) {
                        const float g = sign_get(slo[vi], shi[vi], c0 + k) * gsc;
                        if (g != 0.f) {
                            if (p.v_nw) atomicAdd(gp, g * s.wnw);
                            if (p.v_ne) atomicAdd(gp + 1, g * s.wne);
                            if (p.v_sw) atomicAdd(gp + W, g * s.wsw);
                            if (p.v_se) atomicAdd(gp + W + 1, g * s.wse);
                        }
                        gp += HW;
                    }
                }
                continue;
            }
            const int chs = f.rows * f.RS;
            __syncthreads();                                // tile is all zero here (initial clear / previous flush)
            if (act) {
                // true (unclamped) tap positions: a valid tap lies inside the footprint by construction
                const int o = (s.y0 - f.ylo) * f.RS + (s.x0 - f.xlo);
                for (int k = 0; k < nch; ++k) {
                    const float g = sign_get(slo[vi], shi[vi], c0 + k) * gsc;
                    float *tq = tile + k * chs + o;
                    if (g != 0.f && !DVF_DBG(a, 4)) {
                        if (p.v_nw) atomicAdd(tq, g * s.wnw);
                        if (p.v_ne) atomicAdd(tq + 1, g * s.wne);
                        if (p.v_sw) atomicAdd(tq + f.RS, g * s.wsw);
                        if (p.v_se) atomicAdd(tq + f.RS + 1, g * s.wse);
                    }
                }
            }
            __syncthreads();
            // flush: one row-contiguous global atomic per touched footprint element, and re-zero the tile
            {
                const int total = nch * chs;
                const float inv_rs = 1.0f / (float)f.RS, inv_rows = 1.0f / (float)f.rows;
                float *gbase = gs + ((int64_t)b * C + c0) * HW + (int64_t)f.ylo * W + f.xlo;
                for (int idx = tid; idx < total; idx += 256) {
                    const float val = tile[idx];
                    if (val != 0.f) {
                        const int r = div_small(idx, inv_rs), xx = idx - r * f.RS;
                        const int ch = div_small(r, inv_rows), yy = r - ch * f.rows;
                        if (!DVF_DBG(a, 2)) atomicAdd(gbase + ((int64_t)ch * H + yy) * W + xx, val);
                        tile[idx] = 0.f;
                    }
                }
            }
        }
    }
}

// Second stage of the pose-gradient reduction: pose_ws[(v*B+b)*12 + k] = sum over the blocks of image b, fixed order.
__global__ __launch_bounds__(256) void pose_sum_kernel(const float *part, float *pose_ws, int V, int B, int blocks_per_img) {
    __shared__ float red[21][12];
    const int v = blockIdx.x / B, b = blockIdx.x - v * B;
    const int t = threadIdx.x;
    if (t < 252) {
        const int k = t % 12, j = t / 12;
        const float *p = part + ((int64_t)b * blocks_per_img * V + v) * 12 + k;
        float s = 0.f;
        for (int i = j; i < blocks_per_img; i += 21) s += p[(int64_t)i * V * 12];
        red[j][k] = s;
    }
    __syncthreads();
    if (t < 12) {
        float s = 0.f;
        for (int j = 0; j < 21; ++j) s += red[j][t];
        pose_ws[((int64_t)v * B + b) * 12 + t] = s;
    }
}

inline dim3 photo_grid(int B, int H, int W) { return dim3((W + TX - 1) / TX, (H + TY - 1) / TY, B); }
inline size_t photo_lds(int C) { return (size_t)(C < PT_CC ? C : PT_CC) * PT_CAP * sizeof(float); }

int fill_photo_args(PhotoArgs &a, const float *tgt, const float *const *srcs, int V, const float *depth,
                    const float *pose, const float *K, const float *Kinv, const float *mask, int B, int C,
                    int H, int W, float in_scale, uint32_t flags) {
    if (!tgt || !srcs || !depth || !pose || !K || !Kinv || V < 1 || V > DVF_MAX_VIEWS || B <= 0 || C <= 0 ||
        H < 2 || W < 2 || B > 65535 || !(in_scale != 0.f))
        return DVF_ERR_INVALID_ARG;
    if ((int64_t)C * H * W >= ((int64_t)1 << 31)) return DVF_ERR_UNSUPPORTED;
    a = PhotoArgs{};
    a.tgt = tgt;
    a.vec = (W % 4 == 0) ? 1 : 0;
    for (int v = 0; v < V; ++v) {
        if (!srcs[v]) return DVF_ERR_INVALID_ARG;
        a.src[v] = srcs[v];
        if (reinterpret_cast<uintptr_t>(srcs[v]) & 15) a.vec = 0;
    }
    a.depth = depth; a.pose = pose; a.K = K; a.Kinv = Kinv; a.mask = mask;
    a.B = B; a.C = C; a.H = H; a.W = W; a.V = V;
    a.quat = rot_mode(flags);
    a.in_scale = in_scale;
    if (const char *e = dvf_tune("DVF_PHOTO_DBG")) a.dbg = atoi(e);
    return DVF_OK;
}

}  // namespace

extern "C" {

int64_t dvf_photo_partials_floats(int B, int H, int W, int V) {
    const dim3 g = photo_grid(B, H, W);
    return (int64_t)g.x * g.y * g.z * V;
}

int64_t dvf_photo_pose_ws_floats(int B, int H, int W, int V) {
    const dim3 g = photo_grid(B, H, W);
    return (int64_t)g.x * g.y * g.z * V * 12 + (int64_t)V * B * 12;
}

int dvf_photo_loss_fwd(const float *tgt, const float *const *srcs, int V, const float *depth, const float *pose,
                       const float *K, const float *Kinv, const float *mask, float *loss_out, float *view_loss,
                       float *partials, int B, int C, int H, int W, float in_scale, uint32_t flags, void *stream) {
    PhotoArgs a;
    int rc = fill_photo_args(a, tgt, srcs, V, depth, pose, K, Kinv, mask, B, C, H, W, in_scale, flags);
    if (rc != DVF_OK) return rc;
    if (!loss_out || !partials) return DVF_ERR_INVALID_ARG;
    a.partials = partials;
    hipStream_t st = dvf_stream(stream);
    const dim3 grid = photo_grid(B, H, W);
    const size_t lds = photo_lds(C);
    rc = dispatch_mode(flags, [&](auto border, auto align, auto pix) {
        constexpr bool BD = decltype(border)::value, AL = decltype(align)::value, PX = decltype(pix)::value;
        switch (V) {
            case 1: photo_fwd_kernel<BD, AL, PX, 1><<<grid, dim3(TX, TY), lds, st>>>(a); break;
            case 2: photo_fwd_kernel<BD, AL, PX, 2><<<grid, dim3(TX, TY), lds, st>>>(a); break;
            case 3: photo_fwd_kernel<BD, AL, PX, 3><<<grid, dim3(TX, TY), lds, st>>>(a); break;
            default: photo_fwd_kernel<BD, AL, PX, 4><<<grid, dim3(TX, TY), lds, st>>>(a); break;
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
                       int C, int H, int W, float in_scale, uint32_t flags, void *stream) {
    PhotoArgs a;
    int rc = fill_photo_args(a, tgt, srcs, V, depth, pose, K, Kinv, mask, B, C, H, W, in_scale, flags);
    if (rc != DVF_OK) return rc;
    if (!grad_loss || (g_pose && !pose_ws) || (g_mask && !mask)) return DVF_ERR_INVALID_ARG;
    bool any_src = false;
    for (int v = 0; v < V; ++v) {
        a.g_src[v] = g_srcs ? g_srcs[v] : nullptr;
        any_src |= a.g_src[v] != nullptr;
    }
    if ((g_tgt || any_src) && C > PT_MAXC) return DVF_ERR_UNSUPPORTED;      // (per-channel signs are kept in 64 bits)
    a.grad_loss = grad_loss;
    a.g_depth = g_depth; a.g_tgt = g_tgt; a.g_mask = g_mask;
    const dim3 grid = photo_grid(B, H, W);
    const int64_t nblk = (int64_t)grid.x * grid.y * grid.z;
    a.pose_part = g_pose ? pose_ws + (int64_t)V * B * 12 : nullptr;
    hipStream_t st = dvf_stream(stream);
    const size_t lds = photo_lds(C);
    rc = dispatch_mode(flags, [&](auto border, auto align, auto pix) {
        constexpr bool BD = decltype(border)::value, AL = decltype(align)::value, PX = decltype(pix)::value;
        switch (V) {
            case 1: photo_bwd_kernel<BD, AL, PX, 1><<<grid, dim3(TX, TY), lds, st>>>(a); break;
            case 2: photo_bwd_kernel<BD, AL, PX, 2><<<grid, dim3(TX, TY), lds, st>>>(a); break;
            case 3: photo_bwd_kernel<BD, AL, PX, 3><<<grid, dim3(TX, TY), lds, st>>>(a); break;
            default: photo_bwd_kernel<BD, AL, PX, 4><<<grid, dim3(TX, TY), lds, st>>>(a); break;
        }
        DVF_LAUNCH_CHECK();
        return DVF_OK;
    });
    if (rc != DVF_OK) return rc;
    (void)nblk;
    if (g_pose) {
        pose_sum_kernel<<<V * B, 256, 0, st>>>(a.pose_part, pose_ws, V, B, (int)(grid.x * grid.y));
        DVF_LAUNCH_CHECK();
        pose_finalize_kernel<<<(V * B + 63) / 64, 64, 0, st>>>(pose, pose_ws, g_pose, V * B, rot_mode(flags));
        DVF_LAUNCH_CHECK();
    }
    return DVF_OK;
}

}  // extern "C"
