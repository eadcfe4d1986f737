// Fused inverse-warp + photometric-L1 loss (forward and backward), one launch per pyramid scale for all V views.
//
// A block owns a 64 x 4 tile of TARGET pixels (one pixel per lane, one image row per wave).  Per pixel the
// pixel -> cam -> SE3 -> pixel chain of every view is evaluated in registers (depth and target are read once).  The
// 2x2 source neighbourhoods of a tile's pixels form a slightly larger, displaced tile of the source image: its bounding
// box (the "footprint") is reduced over the block, and
//   * forward / backward pass 1: the footprint of PT_CC channels is STAGED IN LDS with coalesced 16-byte row loads and
//     every pixel reads its four taps from LDS (no per-pixel gathers from HBM);
//   * backward pass 2 (source gradients, the feature-reconstruction loss): the scatter-add of the four tap weights is
//     ACCUMULATED IN THE SAME LDS TILE and flushed once per footprint element with row-contiguous global atomics --
//     four global atomics per (pixel, channel, view) become ~1.3, in 256-byte contiguous shapes.  The LDS accumulation
//     is in 32-bit FIXED POINT (ds_add_u32): on gfx950 ds_add_f32 takes 192 cycles per wave-instruction (three per
//     lane, serialised), ds_add_u32 5-8 (tools/micro/lds_atomic.hip).  Scale per (block, view): 2^22 over the power
//     of two above the largest |upstream value| of the block, so a contribution keeps 22 significant bits of the
//     largest one (fp32 adds keep 24 of the running sum) and the 256 pixels of a block cannot overflow 2^31; integer
//     adds commute, so a block's accumulation is order independent.
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

constexpr int PT_CC = 16;         // channels per staged chunk (at most; fewer when the footprint is large)
constexpr int PT_CAP = 600;       // LDS floats per channel at PT_CC channels: a 68 x 5 footprint (typical) needs 340;
                                  // larger footprints (wild depth: random-init networks) take fewer channels per round
constexpr int PT_MIN_TILE = 4096; // floats: LDS tile of the few-channel (image) launches
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
    int tile_floats;               // size of the dynamic LDS tile
    int caffe;                     // DVF_CAFFE_ABSLOSS: per-sample sum, no exact-zero mask, sign(0) -> one side
    int dbg;                       // ablation switches (-DDVF_TUNING builds only): 1 no pass 2, 2 no flush atomics, 4 no LDS adds, 8 no staging, 16 no LDS tiles at all
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
struct Foot { int xlo, ylo, RS, rows, mode, cc; };   // mode 0: no valid tap, 1: LDS tile (cc channels per round), 2: direct (does not fit)

// Block reduction of one view's tap positions to its footprint (two barriers).
__device__ __forceinline__ Foot footprint(const TapPos &p, int vec, int (*bbs)[4], int tile_floats, bool force_direct) {
    const int mnx = wave_min_i(p.any ? p.xb : 0x7fffffff), mxx = wave_max_i(p.any ? p.xb + 1 : -1);
    const int mny = wave_min_i(p.any ? p.y0 : 0x7fffffff), mxy = wave_max_i(p.any ? p.y1 : -1);
    __syncthreads();                                        // (bbs of the previous view fully read)
    if (threadIdx.x == 0) { bbs[threadIdx.y][0] = mnx; bbs[threadIdx.y][1] = mxx; bbs[threadIdx.y][2] = mny; bbs[threadIdx.y][3] = mxy; }
    __syncthreads();
    int x0 = bbs[0][0], x1 = bbs[0][1], y0 = bbs[0][2], y1 = bbs[0][3];
#pragma unroll
    for (int w = 1; w < TY; ++w) {
        x0 = min(x0, bbs[w][0]); x1 = max(x1, bbs[w][1]);
        y0 = min(y0, bbs[w][2]); y1 = max(y1, bbs[w][3]);
    }
    // (readfirstlane: the values are block-uniform; keeps the staging loops' bookkeeping scalar)
    x0 = __builtin_amdgcn_readfirstlane(x0); x1 = __builtin_amdgcn_readfirstlane(x1);
    y0 = __builtin_amdgcn_readfirstlane(y0); y1 = __builtin_amdgcn_readfirstlane(y1);
    Foot f;
    f.xlo = vec ? (x0 & ~3) : x0;
    f.ylo = y0;
    f.RS = vec ? ((x1 - f.xlo + 1 + 3) & ~3) : (x1 - f.xlo + 1);
    f.rows = y1 - y0 + 1;
    const int fp = f.RS * f.rows;
    f.cc = (x1 < 0 || fp <= 0) ? 0 : min(PT_CC, tile_floats / fp);
    f.mode = (x1 < 0) ? 0 : ((f.cc >= 1 && f.RS <= 256 && !force_direct) ? 1 : 2);
    if (x1 < 0) { f.xlo = f.ylo = 0; f.RS = f.rows = 1; }   // (no valid tap anywhere: callers that still walk the channels read zeros)
    return f;
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

// One view at a time (projection, footprint, staged chunks): only one view's sampling state is live, which keeps the
// kernel at four waves per SIMD -- the tile pipeline is latency bound (barriers) and needs the occupancy.
template <bool BORDER, bool ALIGN, bool PIX, int NV>
__global__ __launch_bounds__(256) void photo_fwd_kernel(PhotoArgs a) {
    __shared__ ViewGeo geo[DVF_MAX_VIEWS];
    __shared__ float kinv[9];
    __shared__ float red[TY][DVF_MAX_VIEWS];
    __shared__ int bbs[TY][4];
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.z, tid = threadIdx.y * TX + threadIdx.x;
    block_setup(a, b, tid, geo, kinv, nullptr);
    const int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
    const int W = a.W, H = a.H, C = a.C;
    const int HW = H * W;
    const bool inside = x < W && y < H;
    const int pix = min(y, H - 1) * W + min(x, W - 1);
    const float d = a.depth[(int64_t)b * HW + pix];
    // cam = (Kinv @ (u, v, 1)) * depth                      inverse_warp.py:38-40
    const float u = (float)x, v = (float)y;
    const float cx = (kinv[0] * u + kinv[1] * v + kinv[2]) * d;
    const float cy = (kinv[3] * u + kinv[4] * v + kinv[5]) * d;
    const float cz = (kinv[6] * u + kinv[7] * v + kinv[8]) * d;
    const float *tg = a.tgt + (int64_t)b * C * HW + pix;
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        const Samp s = project<BORDER, ALIGN, PIX>(geo[vi], cx, cy, cz, W, H);
        const TapPos tp = tap_pos(s, W, H, inside);
        const Foot f = footprint(tp, a.vec, bbs, a.tile_floats, DVF_DBG(a, 16));
        float acc = 0.f;
        bool nz = a.caffe != 0;                             // (Caffe AbsLoss: no exact-zero validity mask)
        if (f.mode != 0 || a.caffe) {                       // (mode 0: no pixel of the block samples inside this source)
            const TileOff o = tile_off(tp, f);
            const int chs = f.rows * f.RS;
            const int cstep = f.mode == 1 ? f.cc : PT_CC;
            for (int c0 = 0; c0 < C; c0 += cstep) {
                const int nch = min(cstep, C - c0);
                if (f.mode == 1) {
                    __syncthreads();                        // previous tile fully consumed
                    if (!DVF_DBG(a, 8)) stage_tile(tile, a.src[vi], f, b, C, c0, nch, H, W, a.in_scale, a.vec, tid);
                    __syncthreads();
                }
                if (!tp.any && !a.caffe) continue;          // warped == 0 in every channel: nz stays false
                const float *sp = a.src[vi] + ((int64_t)b * C + c0) * HW;
#pragma unroll 1
                for (int k0 = 0; k0 < nch; k0 += 4) {       // four channels at a time
                    float tv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        tv[j] = (k0 + j < nch) ? tg[(c0 + k0 + j) * HW] : 0.f;
                        if (a.in_scale != 1.f) tv[j] = __fmul_rn(a.in_scale, tv[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (k0 + j < nch) {
                            const Taps t = (f.mode == 1) ? taps_lds(tile + (k0 + j) * chs, o, tp)
                                                         : taps_direct(sp + (k0 + j) * HW, tp, W, a.in_scale);
                            const float wv = blend(t, s);
                            nz |= (wv != 0.f);              // loss_functions.py:11  (warped == 0).prod(1)
                            acc += fabsf(tv[j] - wv);       // :12-13
                        }
                    }
                }
            }
        }
        float m = 1.f;
        if (a.mask) m = fabsf(a.mask[((int64_t)b * NV + vi) * HW + pix]);      // loss_functions_sfm.py:30-31
        const float r = wave_sum((inside && nz) ? acc * m : 0.f);
        if (threadIdx.x == 0) red[threadIdx.y][vi] = r;
    }
    __syncthreads();
    if (tid < a.V) {
        const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.partials[blk * a.V + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// Deterministic second stage: one block of 1024 threads sums the per-block partials in a fixed order (thread t takes
// blocks t, t + 1024, ... with four loads in flight; then a wave reduction and a fixed-order sum of the 16 wave totals).
__global__ __launch_bounds__(1024) void photo_reduce_kernel(const float *partials, int64_t nblk, int V, float inv_n,
                                                            float *loss_out, float *view_loss) {
    __shared__ float red[16][DVF_MAX_VIEWS];
    float s[DVF_MAX_VIEWS] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i0 = threadIdx.x; i0 < nblk; i0 += 4096) {
        float v[4][DVF_MAX_VIEWS];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < DVF_MAX_VIEWS; ++k) {
                const int64_t i = i0 + 1024 * u;
                v[u][k] = (i < nblk && k < V) ? partials[i * V + k] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < DVF_MAX_VIEWS; ++k) s[k] += v[u][k];
    }
#pragma unroll
    for (int k = 0; k < DVF_MAX_VIEWS; ++k) {
        const float w = wave_sum(s[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int k = 0; k < V; ++k) {
            float t = 0.f;
            for (int w = 0; w < 16; ++w) t += red[w][k];
            t *= inv_n;                                      // mean over B*C*H*W (or / B: DVF_CAFFE_ABSLOSS)
            if (view_loss) view_loss[k] = t;
            total += t;
        }
        loss_out[0] = total;
    }
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
    __shared__ int bbs[TY][4];
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.z, tid = threadIdx.y * TX + threadIdx.x;
    block_setup(a, b, tid, geo, kinv, kmat);
    const int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
    const int W = a.W, H = a.H, C = a.C;
    const int HW = H * W;
    const bool inside = x < W && y < H;
    const int pix = min(y, H - 1) * W + min(x, W - 1);
    const float scale = a.caffe ? a.grad_loss[0] / (float)a.B : a.grad_loss[0] / ((float)a.B * (float)C * (float)H * (float)W);
    const float d = a.depth[(int64_t)b * HW + pix];
    const float u = (float)x, v = (float)y;
    const float c0x = kinv[0] * u + kinv[1] * v + kinv[2];
    const float c0y = kinv[3] * u + kinv[4] * v + kinv[5];
    const float c0z = kinv[6] * u + kinv[7] * v + kinv[8];
    const float cx = c0x * d, cy = c0y * d, cz = c0z * d;
    const float *tg = a.tgt + (int64_t)b * C * HW + pix;
    const bool need_tgt = a.g_tgt != nullptr;
    const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float gd = 0.f;
    float vmv[NV];                                          // validity * explainability * upstream / N per view
    uint32_t slo[NV], shi[NV];                              // L1 signs per (view, channel)
    // ---- pass 1, one view at a time: d loss / d ix, iy (without the validity factor), |diff| sum, exact-zero flag and
    // the L1 signs; then the chain to depth and to the [R|t] gradient partials of that view
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        const Samp s = project<BORDER, ALIGN, PIX>(geo[vi], cx, cy, cz, W, H);
        const TapPos tp = tap_pos(s, W, H, inside);
        const Foot f = footprint(tp, a.vec, bbs, a.tile_floats, DVF_DBG(a, 16));
        const float m = a.mask ? a.mask[((int64_t)b * NV + vi) * HW + pix] : 1.f;
        float gix = 0.f, giy = 0.f, absum = 0.f;
        uint32_t lo = 0u, hi = 0u;
        bool nzf = a.caffe != 0;
        if (f.mode != 0 || a.caffe) {
            const TileOff o = tile_off(tp, f);
            const int chs = f.rows * f.RS;
            const int cstep = f.mode == 1 ? f.cc : PT_CC;
            for (int c0 = 0; c0 < C; c0 += cstep) {
                const int nch = min(cstep, C - c0);
                if (f.mode == 1) {
                    __syncthreads();
                    if (!DVF_DBG(a, 8)) stage_tile(tile, a.src[vi], f, b, C, c0, nch, H, W, a.in_scale, a.vec, tid);
                    __syncthreads();
                }
                if (!tp.any && !a.caffe) continue;
                const float *sp = a.src[vi] + ((int64_t)b * C + c0) * HW;
#pragma unroll 1
                for (int k0 = 0; k0 < nch; k0 += 4) {
                    float tv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        tv[j] = (k0 + j < nch) ? tg[(c0 + k0 + j) * HW] : 0.f;
                        if (a.in_scale != 1.f) tv[j] = __fmul_rn(a.in_scale, tv[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (k0 + j < nch) {
                            const Taps t = (f.mode == 1) ? taps_lds(tile + (k0 + j) * chs, o, tp)
                                                         : taps_direct(sp + (k0 + j) * HW, tp, W, a.in_scale);
                            const float wv = blend(t, s);
                            nzf |= (wv != 0.f);
                            const float df = tv[j] - wv;
                            // sign of the masked difference; Caffe AbsLoss: (warped - tgt > 0) - (warped - tgt <= 0), i.e. a
                            // zero difference counts as tgt >= warped                     abs_loss_layer.cu:31
                            const float sg = a.caffe ? ((df * m >= 0.f) ? 1.f : -1.f) : sgn(df * m);
                            absum += fabsf(df);
                            float dox, doy;
                            blend_grad(t, s, dox, doy);
                            gix -= sg * dox;                // d|.|/d warped = -sign
                            giy -= sg * doy;
                            sign_put(lo, hi, c0 + k0 + j, sg);
                        }
                    }
                }
            }
        }
        slo[vi] = lo; shi[vi] = hi;
        const bool nz = inside && nzf;
        const float vm = nz ? m * scale : 0.f;
        vmv[vi] = vm;
        if (a.g_mask && inside) a.g_mask[((int64_t)b * NV + vi) * HW + pix] = nz ? absum * sgn(m) * scale : 0.f;
        // chain to the projected point                       cam2pixel, inverse_warp.py:61-66
        const float gxq = gix * vm * s.dix, gyq = giy * vm * s.diy;
        const float gpx = gxq / s.Z, gpy = gyq / s.Z;
        const float gpz = s.zpass ? -(gxq * s.xq + gyq * s.yq) / s.Z : 0.f;
        const ViewGeo &g = geo[vi];
        // d p / d depth = (K R) cam0
        const float gcx = g.A[0] * gpx + g.A[3] * gpy + g.A[6] * gpz;
        const float gcy = g.A[1] * gpx + g.A[4] * gpy + g.A[7] * gpz;
        const float gcz = g.A[2] * gpx + g.A[5] * gpy + g.A[8] * gpz;
        gd += nz ? gcx * c0x + gcy * c0y + gcz * c0z : 0.f;
        if (a.pose_part) {
            // y = R cam + t ; g_y = K^T g_p ; g_t and g_R = g_y (x) cam ; reduced over the block and stored per block
            const float gyx = nz ? kmat[0] * gpx + kmat[3] * gpy + kmat[6] * gpz : 0.f;
            const float gyy = nz ? kmat[1] * gpx + kmat[4] * gpy + kmat[7] * gpz : 0.f;
            const float gyz = nz ? kmat[2] * gpx + kmat[5] * gpy + kmat[8] * gpz : 0.f;
            float pacc[1][12] = {{gyx, gyy, gyz, gyx * cx, gyx * cy, gyx * cz, gyy * cx, gyy * cy, gyy * cz, gyz * cx, gyz * cy,
                                  gyz * cz}};
            __syncthreads();                                // (red of the previous view fully read)
            reduce_pose_partials<1>(pacc, 1, a.B, b, nullptr, red, a.pose_part + (blk * NV + vi) * 12);
        }
    }
    if (a.g_depth && inside) a.g_depth[(int64_t)b * HW + pix] = gd;
    // ---- grad target: d loss / d tgt_c = sum over views of sign * validity (x in_scale: the kernel scaled the input)
    if (need_tgt && inside) {
        float *gt = a.g_tgt + (int64_t)b * C * HW + pix;
        for (int c = 0; c < C; ++c) {
            float g = 0.f;
#pragma unroll
            for (int vi = 0; vi < NV; ++vi) g += sign_get(slo[vi], shi[vi], c) * vmv[vi];
            gt[c * HW] = g * a.in_scale;
        }
    }
    bool need_src = false;
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) need_src |= a.g_src[vi] != nullptr;
    if (!need_src || DVF_DBG(a, 1)) return;                 // (block-uniform)
    // ---- pass 2: grad source = scatter-add of -g * (bilinear weights), accumulated per footprint in the LDS tile in
    // fixed point (see the header) and flushed with one row-contiguous global atomic per touched element.  The view's
    // sampling state is recomputed (same arithmetic, same bits) rather than kept alive across pass 1.
    int *itile = reinterpret_cast<int *>(tile);
    {
        typedef int i4 __attribute__((ext_vector_type(4)));
        const int tot4 = a.tile_floats >> 2;
        __syncthreads();                                    // (the last staged tile is still being read by slower waves)
        for (int i = tid; i < tot4; i += 256) reinterpret_cast<i4 *>(itile)[i] = i4{0, 0, 0, 0};
    }
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        float *gs = a.g_src[vi];
        if (!gs) continue;                                  // (block-uniform)
        const Samp s = project<BORDER, ALIGN, PIX>(geo[vi], cx, cy, cz, W, H);
        const TapPos p = tap_pos(s, W, H, inside);
        const Foot f = footprint(p, a.vec, bbs, a.tile_floats, DVF_DBG(a, 16));
        // fixed-point scale from the block's largest |upstream value| (wave max -> LDS -> block max)
        float gmax = p.any ? fabsf(vmv[vi] * a.in_scale) : 0.f;
        if (!(gmax == gmax)) gmax = __builtin_inff();       // NaN upstream: handled like inf (direct path)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
        __syncthreads();
        if (threadIdx.x == 0) bbs[threadIdx.y][0] = __builtin_bit_cast(int, gmax);
        __syncthreads();
        gmax = fmaxf(fmaxf(__builtin_bit_cast(float, bbs[0][0]), __builtin_bit_cast(float, bbs[1][0])),
                     fmaxf(__builtin_bit_cast(float, bbs[2][0]), __builtin_bit_cast(float, bbs[3][0])));
        gmax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, gmax)));
        if (f.mode == 0 || gmax == 0.f) continue;           // nothing to add (block-uniform)
        // S = 2^(22 - e) with 2^(e-1) <= gmax < 2^e (frexp exponent): every |contribution| * S <= 2^22
        int e = 0;
        (void)frexpf(gmax, &e);
        const bool fx_ok = gmax < 3.0e38f && e > -100 && e < 100;
        const float fx_scale = ldexpf(1.f, 22 - e), fx_inv = ldexpf(1.f, e - 22);
        const bool act = p.any && vmv[vi] != 0.f;
        const float gsc = -vmv[vi] * a.in_scale;
        if (f.mode == 2 || !fx_ok) {                        // footprint too large for the tile (or inf/NaN upstream)
            if (act) {
                float *gp = gs + (int64_t)b * C * HW + s.y0 * W + s.x0;
                for (int c = 0; c < C; ++c) {
                    const float g = sign_get(slo[vi], shi[vi], c) * gsc;
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
        // true (unclamped) tap positions: a valid tap lies inside the footprint by construction
        const int o = (s.y0 - f.ylo) * f.RS + (s.x0 - f.xlo);
        const float gq = gsc * fx_scale;                    // exact: power-of-two scale
        const int inw = p.v_nw ? __float2int_rn(gq * s.wnw) : 0, ine = p.v_ne ? __float2int_rn(gq * s.wne) : 0;
        const int isw = p.v_sw ? __float2int_rn(gq * s.wsw) : 0, ise = p.v_se ? __float2int_rn(gq * s.wse) : 0;
        for (int c0 = 0; c0 < C; c0 += f.cc) {
            const int nch = min(f.cc, C - c0);
            __syncthreads();                                // tile is all zero here (initial clear / previous flush)
            if (act && !DVF_DBG(a, 4)) {
                for (int k = 0; k < nch; ++k) {
                    const float sg = sign_get(slo[vi], shi[vi], c0 + k);
                    int *tq = itile + k * chs + o;
                    if (sg != 0.f) {
                        const bool neg = sg < 0.f;          // (-x rounds to -(rint x): the sign can be applied after)
                        if (p.v_nw) atomicAdd(tq, neg ? -inw : inw);
                        if (p.v_ne) atomicAdd(tq + 1, neg ? -ine : ine);
                        if (p.v_sw) atomicAdd(tq + f.RS, neg ? -isw : isw);
                        if (p.v_se) atomicAdd(tq + f.RS + 1, neg ? -ise : ise);
                    }
                }
            }
            __syncthreads();
            // flush: one row-contiguous global atomic per touched footprint element, and re-zero the tile
            const int total = nch * chs;
            const float inv_rs = 1.0f / (float)f.RS, inv_rows = 1.0f / (float)f.rows;
            float *gbase = gs + ((int64_t)b * C + c0) * HW + f.ylo * W + f.xlo;
            for (int idx = tid; idx < total; idx += 256) {
                const int val = itile[idx];
                if (val != 0) {
                    const int r = div_small(idx, inv_rs), xx = idx - r * f.RS;
                    const int ch = div_small(r, inv_rows), yy = r - ch * f.rows;
                    if (!DVF_DBG(a, 2)) atomicAdd(gbase + (ch * H + yy) * W + xx, (float)val * fx_inv);
                    itile[idx] = 0;
                }
            }
        }
    }
}

// Second stage of the pose-gradient reduction: pose_ws[(v*B+b)*12 + k] = sum over the blocks of image b, fixed order
// (1008 threads = 84 strided partial sums per value, then a fixed-order sum of the 84).
__global__ __launch_bounds__(1024) void pose_sum_kernel(const float *part, float *pose_ws, int V, int B, int blocks_per_img) {
    __shared__ float red[84][12];
    const int v = blockIdx.x / B, b = blockIdx.x - v * B;
    const int t = threadIdx.x;
    if (t < 1008) {
        const int k = t % 12, j = t / 12;
        const float *p = part + ((int64_t)b * blocks_per_img * V + v) * 12 + k;
        float s0 = 0.f, s1 = 0.f;
        int i = j;
        for (; i + 84 < blocks_per_img; i += 168) {          // two loads in flight
            const float a0 = p[(int64_t)i * V * 12], a1 = p[(int64_t)(i + 84) * V * 12];
            s0 += a0;
            s1 += a1;
        }
        if (i < blocks_per_img) s0 += p[(int64_t)i * V * 12];
        red[j][k] = s0 + s1;
    }
    __syncthreads();
    if (t < 12) {
        float s = 0.f;
        for (int j = 0; j < 84; ++j) s += red[j][t];
        pose_ws[((int64_t)v * B + b) * 12 + t] = s;
    }
}

inline dim3 photo_grid(int B, int H, int W) { return dim3((W + TX - 1) / TX, (H + TY - 1) / TY, B); }
inline int photo_tile_floats(int C) { const int t = (C < PT_CC ? C : PT_CC) * PT_CAP; return t < PT_MIN_TILE ? PT_MIN_TILE : t; }

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
    a.caffe = (flags & DVF_CAFFE_ABSLOSS) ? 1 : 0;
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
    a.tile_floats = photo_tile_floats(C);
    const size_t lds = (size_t)a.tile_floats * sizeof(float);
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
    const float inv_n = (flags & DVF_CAFFE_ABSLOSS) ? 1.f / (float)B : 1.f / ((float)B * (float)C * (float)H * (float)W);
    photo_reduce_kernel<<<1, 1024, 0, st>>>(partials, (int64_t)grid.x * grid.y * grid.z, V, inv_n, loss_out, view_loss);
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
    a.tile_floats = photo_tile_floats(C);
    const size_t lds = (size_t)a.tile_floats * sizeof(float);
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
        pose_sum_kernel<<<V * B, 1024, 0, st>>>(a.pose_part, pose_ws, V, B, (int)(grid.x * grid.y));
        DVF_LAUNCH_CHECK();
        pose_finalize_kernel<<<(V * B + 63) / 64, 64, 0, st>>>(pose, pose_ws, g_pose, V * B, rot_mode(flags));
        DVF_LAUNCH_CHECK();
    }
    return DVF_OK;
}

}  // extern "C"
