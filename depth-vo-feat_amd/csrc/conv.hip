// fp32 MFMA convolution kernels for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, bit-identical to an fmaf
// chain, so the 1e-4 parity bound holds without any reduced-precision path).
//
// Two kernels cover every convolution-shaped op of DispNetS / PoseExpNet / FeatExtractor:
//
//  conv_gather_kernel   out[m][Y][X] = sum_r sum_taps W(m, r, tap) * in[r][iy][ix]      ("gather form")
//      GEMM view: M = output channels (MFMA rows, 32 per tile), N = 32 output pixels per tile (one pixel per
//      lane -> coalesced NCHW stores), K = (input channel pair) x taps.  The input patch (+halo) of CK channels
//      and the matching weight slab are staged in LDS once per chunk and re-used by every tap.
//      Used as: Conv2d forward (stride 1/2), Conv2d dgrad and ConvTranspose2d forward (one launch per output
//      parity class, so strided layers waste no MACs on zeros), ConvTranspose2d dgrad.  Virtual concat: up to 3
//      input segments are walked in place of torch.cat.
//
//  conv_wgrad_kernel    G[m][c][tap] += sum_pixels P[m][pix] * Q[c][pix*S + tap - pad]
//      GEMM view: M = P channels, N = 32 (c, tap) columns per tile, K = pixel pairs.  Conv2d: P = dL/dpre,
//      Q = input;  ConvTranspose2d: P = input, Q = dL/dpre.  Each block keeps its G tile in MFMA accumulators
//      across many pixel tiles and adds it to global memory once.
//
// Layouts are the reference's (NCHW activations, OIHW / IOHW weights): nothing is repacked between steps.
#include "dvf_common.h"
#include <string.h>
#include "conv_pipe.h"
#include "conv_head.h"
#include "wgrad_pipe.h"

namespace {

using dvfp::f32x16;
using dvfp::tensor_rsrc;
using dvfp::OOB;
using dvfp::bload;
using dvfp::apply_act;

struct ClassDev { int py, px, by, bx, TA, TB, OHc, OWc, tap0; };

struct GatherArgs {
    const float *in[DVF_MAX_SEGS];
    int segC[DVF_MAX_SEGS];
    int nseg;
    const float *w;
    int w_mode;          // 0: w[((m_base+m)*Rtot + r)*KK + tap]   1: w[(r*Mtot + m_base+m)*KK + tap]
    int Mtot, Rtot, KK, m_base, M;
    const float *bias;   // indexed by m (already offset by the caller) or NULL
    float *out;          // [N, M, OH, OW]
    int N, IH, IW, OH, OW;
    int OS, IS;          // output pixel (oy*OS+py, ox*OS+px); input pixel (oy*IS+by+ta, ox*IS+bx+tb)
    int ncls;            // output-parity classes handled by this launch (blockIdx.z % ncls)
    ClassDev cls[4];
    int act;
    float alpha, beta;
    int KS, NCH, atomic_out, dbg;        // atomic_out 0: bias+act store, 1: atomicAdd into a zeroed output, 2: plain partial store at ks*ws_slice
    int64_t ws_slice;
    int lsw, lsh, TGX, BW, BH, tilesX, tilesY;
    int PSmax, Tmax, WD; // LDS carve: patch CK*PSmax | weights Tmax*CK*COTP | wdec WD | tapB Tmax | inv 64
    int tapmap[52];      // class c uses tapmap[cls[c].tap0 + t]: class tap -> tap index of the stored kernel
    const float *mask;   // (dgrad, atomic_out 0) ReLU backward of the producing layer: out = (mask > 0) ? v : 0; same shape as out
    int *mask_done;      // host side: set to 1 when the launch applied the mask itself
};

// MT x 32 output channels, 4*NT tiles of 32 pixels per block, CK = 2*CKH reduction channels per LDS chunk.
template <int MT, int NT, int CKH>
__global__ __launch_bounds__(256) void conv_gather_kernel(const GatherArgs a) {
    constexpr int CK = 2 * CKH, COTP = 32 * MT + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                                   // [CK][PH][RS] (IS==2: columns split by parity)
    float *wl = smem + CK * a.PSmax;                       // [T][CK][COTP]
    int *wdec = reinterpret_cast<int *>(wl + a.Tmax * CK * COTP);   // weight run element -> LDS offset
    int *tapB = wdec + a.WD;                               // class tap -> patch offset
    int *inv = tapB + a.Tmax;                              // stored tap -> class tap
    const int tid = threadIdx.x, lane = tid & 63, nl = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: row/column bookkeeping stays scalar
    const int zc = blockIdx.z % a.ncls, zr = blockIdx.z / a.ncls;
    const int n = zr / a.KS, ks = zr - n * a.KS;
    const ClassDev c = a.cls[zc];
    const int tX = blockIdx.x % a.tilesX, tY = blockIdx.x / a.tilesX;
    const int oy0 = tY * a.BH, ox0 = tX * a.BW;
    if (oy0 >= c.OHc || ox0 >= c.OWc) return;              // tile outside this (smaller) class: whole block exits
    const int m0 = blockIdx.y * (32 * MT);
    const int SW = 1 << a.lsw, SH = 1 << a.lsh;
    const int pxl = nl & (SW - 1), pyl = nl >> a.lsw;
    const int T = c.TA * c.TB;
    const int PH = (a.BH - 1) * a.IS + c.TA, PW = (a.BW - 1) * a.IS + c.TB, PWH = (PW + 1) >> 1;
    const int RS = (a.IS == 2) ? 2 * PWH : PW, PS = PH * RS;

    // ---- one-time tables
    for (int e = tid; e < 64; e += 256) inv[e] = -1;
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        inv[a.tapmap[c.tap0 + t]] = t;
        const int ta = t / c.TB, tb = t - ta * c.TB;
        tapB[t] = ta * RS + ((a.IS == 2) ? ((tb & 1) * PWH + (tb >> 1)) : tb);
    }
    __syncthreads();
    {
        const int nE = (a.w_mode == 0 ? CK : 32 * MT) * a.KK;
        for (int e = tid; e < nE; e += 256) {
            const int hi = e / a.KK, t = inv[e - hi * a.KK];        // hi = r (mode 0) or m (mode 1)
            int d = -1;
            if (t >= 0) d = (hi << 24) | (a.w_mode == 0 ? (t * CK + hi) * COTP : t * CK * COTP + hi);
            wdec[e] = d;
        }
    }

    int boff[NT], opy[NT], opx[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int q = wave * NT + i, tx = q % a.TGX, ty = q / a.TGX;
        opy[i] = ty * SH + pyl;
        opx[i] = tx * SW + pxl;
        boff[i] = kh * PS + (opy[i] * a.IS) * RS + opx[i];
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;

    const int iy0 = oy0 * a.IS + c.by, ix0 = ox0 * a.IS + c.bx;
    // lane-invariant column part of the patch staging: byte offset inside a row (or OOB) and LDS column (or -1)
    const bool npass2 = PW > 64;
    unsigned pcol_off[2];
    int pcol_dst[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int cc = lane + 64 * ps, ix = ix0 + cc;
        pcol_off[ps] = (cc < PW && ix >= 0 && ix < a.IW) ? ((unsigned)ix << 2) : OOB;
        pcol_dst[ps] = (cc < PW) ? ((a.IS == 2) ? ((cc & 1) * PWH + (cc >> 1)) : cc) : -1;
    }
    const __amdgpu_buffer_rsrc_t rs_w = tensor_rsrc(a.w);
    const int g_begin = (int)(((int64_t)a.NCH * ks) / a.KS), g_end = (int)(((int64_t)a.NCH * (ks + 1)) / a.KS);
    int seg = 0, seg_first = 0, r_seg = 0;     // r_seg: index of the segment's first channel in the concat
    for (int g = g_begin; g < g_end; ++g) {
        while (true) {
            const int nchs = (a.segC[seg] + CK - 1) / CK;
            if (g < seg_first + nchs) break;
            seg_first += nchs;
            r_seg += a.segC[seg];
            ++seg;
        }
        const int c0 = (g - seg_first) * CK;                  // first channel of the chunk inside its segment
        const int nch = min(CK, a.segC[seg] - c0);            // valid channels in this chunk
        __syncthreads();                                      // previous chunk fully consumed (and tables visible)
        // ---- stage the input patch.  Row bookkeeping is scalar and incremental (no divisions); everything that
        // depends on the lane (column offset, column validity, LDS column) was computed once before the chunk loop.
        // 8 independent buffer loads are issued before the 8 LDS stores.
        if (!DVF_DBG(a, 1)) {
            const __amdgpu_buffer_rsrc_t rs_in = tensor_rsrc(a.in[seg]);
            const unsigned cb = (unsigned)((n * a.segC[seg] + c0) * a.IH * a.IW) << 2;      // chunk base, bytes
            int ci = 0, r = wave;                              // this wave's next row (ci, r); rows advance by 4
            while (r >= PH) { r -= PH; ++ci; }
            while (ci < CK) {
                float v[8][2];
                int dsts[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dsts[k] = -1;
                    if (ci < CK) {
                        const int iy = iy0 + r;
                        const bool rowok = (ci < nch) & (iy >= 0) & (iy < a.IH);
                        const unsigned soff = cb + ((unsigned)((ci * a.IH + iy) * a.IW) << 2);
                        dsts[k] = ci * PS + r * RS;
                        v[k][0] = bload(rs_in, rowok ? pcol_off[0] : OOB, rowok ? soff : 0u);
                        if (npass2) v[k][1] = bload(rs_in, rowok ? pcol_off[1] : OOB, rowok ? soff : 0u);
                        r += 4;
                        while (r >= PH) { r -= PH; ++ci; }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (dsts[k] >= 0) {
                        if (pcol_dst[0] >= 0) patch[dsts[k] + pcol_dst[0]] = v[k][0];
                        if (npass2 && pcol_dst[1] >= 0) patch[dsts[k] + pcol_dst[1]] = v[k][1];
                    }
                }
            }
        }
        // ---- stage the weight slab wl[(t*CK + r)*COTP + m]: lanes walk the contiguous run of the stored tensor
        // (coalesced), wdec[] maps a run position to its LDS slot; row bases are scalar 32-bit offsets.
        if (!DVF_DBG(a, 2)) {
            const int r0 = r_seg + c0;                        // reduction index of the chunk's first channel
            if (a.w_mode == 0) {
                const int nE = CK * a.KK;
                const unsigned wb = (unsigned)(((a.m_base + m0) * a.Rtot + r0) * a.KK) << 2;
                const unsigned rowstep = (unsigned)(a.Rtot * a.KK) << 2;
                for (int e = lane; e < nE; e += 64) {
                    const int d = wdec[e];
                    const bool use = d >= 0;
                    const unsigned voff = (use && ((d >> 24) < nch)) ? ((unsigned)e << 2) : OOB;
                    const int off = d & 0xFFFFFF;
                    float v[8 * MT];
#pragma unroll
                    for (int k = 0; k < 8 * MT; ++k) {
                        const int m = wave + 4 * k;
                        const bool mok = m0 + m < a.M;
                        v[k] = bload(rs_w, mok ? voff : OOB, mok ? wb + (unsigned)m * rowstep : 0u);
                    }
                    if (use) {
#pragma unroll
                        for (int k = 0; k < 8 * MT; ++k) wl[off + wave + 4 * k] = v[k];
                    }
                }
            } else {
                const int nE = 32 * MT * a.KK;
                const unsigned wb = (unsigned)((r0 * a.Mtot + a.m_base + m0) * a.KK) << 2;
                const unsigned rowstep = (unsigned)(a.Mtot * a.KK) << 2;
                constexpr int RK = (CK + 3) / 4;                // reduction rows per wave
                for (int e0 = lane; e0 < nE; e0 += 256) {
                    int d[4];
                    float v[4][RK];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int e = e0 + 64 * jj;
                        d[jj] = (e < nE) ? wdec[e] : -1;
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int e = e0 + 64 * jj;
                        const unsigned voff = ((d[jj] >= 0) && (m0 + (d[jj] >> 24) < a.M)) ? ((unsigned)e << 2) : OOB;
#pragma unroll
                        for (int k = 0; k < RK; ++k) {
                            const int rr = wave + 4 * k;
                            const bool rok = rr < nch;
                            v[jj][k] = bload(rs_w, rok ? voff : OOB, rok ? wb + (unsigned)rr * rowstep : 0u);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (d[jj] < 0) continue;
#pragma unroll
                        for (int k = 0; k < RK; ++k) {
                            const int rr = wave + 4 * k;
                            if (rr < CK) wl[(d[jj] & 0xFFFFFF) + rr * COTP] = v[jj][k];
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- MFMA: per tap, all CKH channel pairs; the next tap's fragments are fetched before this tap's MFMAs
        if (!DVF_DBG(a, 4)) {
            float af[2][CKH][MT], bf[2][CKH][NT];
            auto load = [&](auto bufc, int t) {
                constexpr int buf = decltype(bufc)::value;
                const int tb_off = tapB[t];
                const int wo = (t * CK + kh) * COTP + nl;
#pragma unroll
                for (int cp = 0; cp < CKH; ++cp) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) af[buf][cp][m] = wl[wo + cp * 2 * COTP + m * 32];
#pragma unroll
                    for (int i = 0; i < NT; ++i) bf[buf][cp][i] = patch[boff[i] + tb_off + cp * 2 * PS];
                }
            };
            auto mma = [&](auto bufc) {
                constexpr int buf = decltype(bufc)::value;
#pragma unroll
                for (int cp = 0; cp < CKH; ++cp)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < NT; ++i)
                            acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][cp][m], bf[buf][cp][i], acc[m][i], 0, 0, 0);
            };
            using B0 = std::integral_constant<int, 0>;
            using B1 = std::integral_constant<int, 1>;
            load(B0{}, 0);
            for (int t = 0; t < T; t += 2) {
                if (t + 1 < T) load(B1{}, t + 1);
                mma(B0{});
                if (t + 1 < T) {
                    if (t + 2 < T) load(B0{}, t + 2);
                    mma(B1{});
                }
            }
        }
    }
    // ---- epilogue: D[row = channel][col = pixel]; lanes 0-31 / 32-63 hold channel rows +0 / +4.  Output mode, activation
    // and "every channel of the block exists" are launch constants: one store loop per combination (per element they cost
    // a chain of scalar branches, an inlined sigmoid, three 64-bit multiplies and a dependent bias load each).
    const int64_t HWo = (int64_t)a.OH * a.OW;
    const bool full_m = m0 + 32 * MT <= a.M;
    auto store_all = [&](auto modec, auto actc, auto fullc) __attribute__((always_inline)) {
        constexpr int MODE = decltype(modec)::value, ACT = decltype(actc)::value;
        constexpr bool FULL = decltype(fullc)::value;
        float bv[MT][16];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = m0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    bv[m][r] = (a.bias && mm < a.M) ? a.bias[mm] : 0.f;        // (all loads in flight together)
                }
        }
        float *obase = a.out + (MODE == 2 ? (int64_t)ks * a.ws_slice : 0) + ((int64_t)n * a.M + m0 + 4 * kh) * HWo;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int oy = oy0 + opy[i], ox = ox0 + opx[i];
            const int Y = oy * a.OS + c.py, X = ox * a.OS + c.px;
            const bool pok = (oy < c.OHc) && (ox < c.OWc) && (Y < a.OH) && (X < a.OW);
            float *pbase = obase + (int64_t)Y * a.OW + X;
            float mk[MT][16];
            if constexpr (MODE == 3) {           // all mask loads of the tile first (invalid elements read element 0)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = m * 32 + (r & 3) + 8 * (r >> 2);
                        const bool ok = pok && (FULL || m0 + 4 * kh + ml < a.M);
                        mk[m][r] = a.mask[ok ? (pbase - a.out) + (int64_t)ml * HWo : 0];
                    }
            }
            if (pok) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = m * 32 + (r & 3) + 8 * (r >> 2);
                        if (FULL || m0 + 4 * kh + ml < a.M) {
                            float *op = pbase + (int64_t)ml * HWo;          // (ml * HWo: scalar)
                            const float v = acc[m][i][r];
                            if constexpr (MODE == 3) {                   // dgrad + ReLU backward of the producing layer
                                *op = mk[m][r] > 0.f ? v : 0.f;
                            } else if constexpr (MODE == 0) {
                                const float t = v + bv[m][r];
                                if constexpr (ACT == DVF_ACT_RELU) *op = fmaxf(t, 0.f);
                                else if constexpr (ACT == DVF_ACT_SIGMOID_AFFINE) *op = a.alpha * (1.f / (1.f + expf(-t))) + a.beta;
                                else *op = t;
                            } else if constexpr (MODE == 1) {
                                atomicAdd(op, v);
                            } else {
                                *op = v;
                            }
                        }
                    }
                }
            }
        }
    };
    using std::integral_constant;
    auto by_full = [&](auto modec, auto actc) __attribute__((always_inline)) {
        if (full_m) store_all(modec, actc, std::true_type{});
        else store_all(modec, actc, std::false_type{});
    };
    if (a.atomic_out == 0 && a.mask) by_full(integral_constant<int, 3>{}, integral_constant<int, DVF_ACT_NONE>{});
    else if (a.atomic_out == 2) by_full(integral_constant<int, 2>{}, integral_constant<int, DVF_ACT_NONE>{});
    else if (a.atomic_out == 1) by_full(integral_constant<int, 1>{}, integral_constant<int, DVF_ACT_NONE>{});
    else if (a.act == DVF_ACT_RELU) by_full(integral_constant<int, 0>{}, integral_constant<int, DVF_ACT_RELU>{});
    else if (a.act == DVF_ACT_SIGMOID_AFFINE) by_full(integral_constant<int, 0>{}, integral_constant<int, DVF_ACT_SIGMOID_AFFINE>{});
    else by_full(integral_constant<int, 0>{}, integral_constant<int, DVF_ACT_NONE>{});
}

// y = act(y + bias[c]) in place: finishes a split-K convolution.
__global__ void bias_act_kernel(float *y, const float *bias, int C, int64_t HW, int64_t total, int act, float alpha,
                                float beta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        float v = y[i];
        if (bias) v += bias[c];
        y[i] = apply_act(v, act, alpha, beta);
    }
}

// ------------------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
    const float *P;      // [N, PCtot, GH, GW]; rows m_base .. m_base+M-1 are used
    const float *Q;      // [N, QCtot, QH, QW]; channels q_base .. q_base+Cq-1 are used
    float *G;            // G[(m_base_g + m) * g_mstride + (g_cbase + c) * KK + tap]
    int PCtot, m_base, M, QCtot, q_base, Cq;
    int64_t g_mstride;
    int g_mbase, g_cbase, KK, KH, KW;
    int N, GH, GW, QH, QW, S, pad;
    int CK, BH, lnp, tilesX, tilesY, PSPLIT;
    int PHq, PWq, PWH, RS, PS, COTP;
    int vp;              // P tile loaded in 16-byte lanes (GW % 4 == 0, 16-byte aligned base)
    int dbg;             // ablation switches (-DDVF_TUNING builds): 1 no Q loads, 2 no P loads, 4 no MFMA, 8 no atomic epilogue, 16 no LDS stores
};

constexpr int WG_BW = 32;

template <int MT, int NTW, bool PF>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int COTP = 32 * MT + 1;              // padded row of the transposed P tile (compile-time: immediate offsets)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *qp = smem;                               // [CK][PHq][RS]
    float *pl = smem + a.CK * a.PS;                 // [BH*32 pixels][COTP]
    const int tid = threadIdx.x, lane = tid & 63, nl = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * (32 * MT);
    const int c0 = blockIdx.y * a.CK;
    const int nch = min(a.CK, a.Cq - c0);
    const int T = a.KH * a.KW;
    const int ncols = nch * T;
    // this lane's columns: tile u of this wave covers columns (wave*NTW+u)*32 + nl
    int loff[NTW], gcol[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int j = (wave * NTW + u) * 32 + nl;
        if (j < ncols) {
            const int cj = j / T, tj = j - cj * T, ta = tj / a.KW, tb = tj - ta * a.KW;
            const int toff = (a.S == 2) ? ((tb & 1) * a.PWH + (tb >> 1)) : tb;
            loff[u] = cj * a.PS + ta * a.RS + toff;
            gcol[u] = (a.g_cbase + c0 + cj) * a.KK + tj;
        } else {
            loff[u] = 0;
            gcol[u] = -1;
        }
    }
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][u][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rs_q = tensor_rsrc(a.Q), rs_p = tensor_rsrc(a.P);
    const bool qpass2 = a.PWq > 64;
    int qcol_dst[2];                               // lane-invariant LDS column of the Q patch (or -1)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int cc = lane + 64 * ps;
        qcol_dst[ps] = (cc < a.PWq) ? ((a.S == 2) ? ((cc & 1) * a.PWH + (cc >> 1)) : cc) : -1;
    }
    const int ntiles = a.N * a.tilesX * a.tilesY;
    // Staging is split into ISSUE (global loads into registers) and STORE (registers -> LDS).  With PF the loads of
    // tile t+1 are issued right before the MFMA phase of tile t and stored after it, so their latency hides behind the
    // MFMAs (the register budget allows it when a wave stages at most WG_QMAX patch rows in one pass).
    constexpr int WG_QMAX = 24, WG_PMAX = 16 * MT;
    float qreg[WG_QMAX], preg[WG_PMAX];
    const int npairs = a.BH >> 1, ptotal = 32 * MT * npairs;
    auto tile_origin = [&](int tile, int &n, int &gy0, int &gx0) {
        n = tile / (a.tilesX * a.tilesY);
        const int rem = tile - n * (a.tilesX * a.tilesY);
        const int tY = rem / a.tilesX, tX = rem - tY * a.tilesX;
        gy0 = tY * a.BH;
        gx0 = tX * WG_BW;
    };
    auto issue_tile = [&](int tile) {
        int n, gy0, gx0;
        tile_origin(tile, n, gy0, gx0);
        if (!DVF_DBG(a, 1)) {   // Q patch rows wave, wave+4, ...: one 64-lane load per row (single pass: PWq <= 64)
            const int qy0 = gy0 * a.S - a.pad, qx0 = gx0 * a.S - a.pad;
            const unsigned cb = (unsigned)((n * a.QCtot + a.q_base + c0) * a.QH * a.QW) << 2;
            const int ix = qx0 + lane;
            const unsigned qoff = (lane < a.PWq && ix >= 0 && ix < a.QW) ? ((unsigned)ix << 2) : OOB;
            int ci = 0, r = wave;
            while (r >= a.PHq) { r -= a.PHq; ++ci; }
#pragma unroll
            for (int k = 0; k < WG_QMAX; ++k) {
                if (ci < a.CK) {
                    const int iy = qy0 + r;
                    const bool rowok = (ci < nch) & (iy >= 0) & (iy < a.QH);
                    const unsigned soff = cb + ((unsigned)((ci * a.QH + iy) * a.QW) << 2);
                    qreg[k] = bload(rs_q, rowok ? qoff : OOB, rowok ? soff : 0u);
                    r += 4;
                    while (r >= a.PHq) { r -= a.PHq; ++ci; }
                }
            }
        }
        if (DVF_DBG(a, 2)) {
        } else if (a.vp) {
            // P tile in 16-byte lanes: instruction k of this wave = image row (k' % BH) of the 8 channels 8*(k'/BH)..+7,
            // k' = wave + 4k; lane = (channel seg = lane >> 3, pixel group q = lane & 7 -> pixels 4q..4q+3).  4x fewer
            // vector-memory instructions than one float per lane: this kernel is bound by their issue rate.
            const int seg = lane >> 3, q = lane & 7;
            const unsigned plane = (unsigned)(a.GH * a.GW) << 2;
            const bool colok = gx0 + 4 * q < a.GW;                       // (GW % 4 == 0: a group is all in or all out)
            const unsigned pb = (unsigned)((n * a.PCtot + a.m_base + m0) * a.GH * a.GW) << 2;
            const unsigned lane_off = (unsigned)seg * plane + ((unsigned)(gx0 + 4 * q) << 2);
            const int nitems = 4 * MT * a.BH;
#pragma unroll
            for (int k = 0; k < WG_PMAX / 4; ++k) {
                const int it = wave + 4 * k, mg = it / a.BH, row = it - mg * a.BH;
                const int gy = gy0 + row;
                const bool ok = (it < nitems) & colok & (gy < a.GH) & (m0 + mg * 8 + seg < a.M);
                const unsigned soff = pb + (unsigned)(mg * 8) * plane + ((unsigned)(gy * a.GW) << 2);
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 v4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, ok ? lane_off : OOB, ok ? soff : 0u, 0));
                preg[4 * k + 0] = v4.x; preg[4 * k + 1] = v4.y; preg[4 * k + 2] = v4.z; preg[4 * k + 3] = v4.w;
            }
        } else {   // P tile: item idx = (channel m, row pair rp); half-wave = one row of 32 pixels
            const int px = lane & 31, prow = lane >> 5;
            const bool colok = gx0 + px < a.GW;
            const unsigned pb = (unsigned)((n * a.PCtot + a.m_base + m0) * a.GH * a.GW) << 2;
#pragma unroll
            for (int k = 0; k < WG_PMAX; ++k) {
                const int idx = wave + 4 * k;
                const int m = idx >> a.lnp, rp = idx - (m << a.lnp);
                const int gy = gy0 + rp * 2 + prow;
                const bool mok = (idx < ptotal) & (m0 + m < a.M);
                const unsigned voff = (colok && gy < a.GH) ? ((unsigned)(gy * a.GW + gx0 + px) << 2) : OOB;
                preg[k] = bload(rs_p, mok ? voff : OOB, mok ? pb + ((unsigned)(m * a.GH * a.GW) << 2) : 0u);
            }
        }
    };
    auto store_tile = [&]() {
        if (DVF_DBG(a, 16)) return;
        {
            int ci = 0, r = wave;
            while (r >= a.PHq) { r -= a.PHq; ++ci; }
#pragma unroll
            for (int k = 0; k < WG_QMAX; ++k) {
                if (ci < a.CK) {
                    if (qcol_dst[0] >= 0) qp[ci * a.PS + r * a.RS + qcol_dst[0]] = qreg[k];
                    r += 4;
                    while (r >= a.PHq) { r -= a.PHq; ++ci; }
                }
            }
        }
        if (a.vp) {
            const int seg = lane >> 3, q = lane & 7;
            const int nitems = 4 * MT * a.BH;
#pragma unroll
            for (int k = 0; k < WG_PMAX / 4; ++k) {
                const int it = wave + 4 * k, mg = it / a.BH, row = it - mg * a.BH;
                if (it < nitems) {
                    float *dst = pl + (row * WG_BW + 4 * q) * COTP + mg * 8 + seg;     // bank = 4q + j + seg: conflict-free
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[j * COTP] = preg[4 * k + j];
                }
            }
        } else {
            const int px = lane & 31, prow = lane >> 5;
#pragma unroll
            for (int k = 0; k < WG_PMAX; ++k) {
                const int idx = wave + 4 * k;
                const int m = idx >> a.lnp, rp = idx - (m << a.lnp);
                if (idx < ptotal) pl[((rp * 2 + prow) * WG_BW + px) * COTP + m] = preg[k];
            }
        }
    };
    if (PF && blockIdx.z < ntiles) issue_tile(blockIdx.z);
    for (int tile = blockIdx.z; tile < ntiles; tile += a.PSPLIT) {
        int n, gy0, gx0;
        tile_origin(tile, n, gy0, gx0);
        __syncthreads();
        if (PF) {
            store_tile();
        } else {
        // ---- stage Q patch: scalar incremental row bookkeeping, lane-invariant column part, buffer loads (zero fill)
        {
            const int qy0 = gy0 * a.S - a.pad, qx0 = gx0 * a.S - a.pad;
            const unsigned cb = (unsigned)((n * a.QCtot + a.q_base + c0) * a.QH * a.QW) << 2;
            unsigned qcol_off[2];
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int cc = lane + 64 * ps, ix = qx0 + cc;
                qcol_off[ps] = (cc < a.PWq && ix >= 0 && ix < a.QW) ? ((unsigned)ix << 2) : OOB;
            }
            int ci = 0, r = wave;
            while (r >= a.PHq) { r -= a.PHq; ++ci; }
            while (ci < a.CK) {
                float v[8][2];
                int dsts[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dsts[k] = -1;
                    if (ci < a.CK) {
                        const int iy = qy0 + r;
                        const bool rowok = (ci < nch) & (iy >= 0) & (iy < a.QH);
                        const unsigned soff = cb + ((unsigned)((ci * a.QH + iy) * a.QW) << 2);
                        dsts[k] = ci * a.PS + r * a.RS;
                        v[k][0] = bload(rs_q, rowok ? qcol_off[0] : OOB, rowok ? soff : 0u);
                        if (qpass2) v[k][1] = bload(rs_q, rowok ? qcol_off[1] : OOB, rowok ? soff : 0u);
                        r += 4;
                        while (r >= a.PHq) { r -= a.PHq; ++ci; }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (dsts[k] >= 0) {
                        if (qcol_dst[0] >= 0) qp[dsts[k] + qcol_dst[0]] = v[k][0];
                        if (qpass2 && qcol_dst[1] >= 0) qp[dsts[k] + qcol_dst[1]] = v[k][1];
                    }
                }
            }
        }
        // ---- stage P tile transposed: pl[pixel][m]; half-wave = one row of 32 pixels; 8 buffer loads per batch
        {
            const int npairs = a.BH >> 1;          // row pairs per channel
            const int px = lane & 31, prow = lane >> 5;
            const int total = 32 * MT * npairs;
            const bool colok = gx0 + px < a.GW;
            const unsigned pb = (unsigned)((n * a.PCtot + a.m_base + m0) * a.GH * a.GW) << 2;
            for (int ib = wave; ib < total; ib += 32) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int idx = ib + 4 * k;
                    const int m = idx >> a.lnp, rp = idx - (m << a.lnp);
                    const int gy = gy0 + rp * 2 + prow;
                    const bool mok = (idx < total) & (m0 + m < a.M);
                    const unsigned voff = (colok && gy < a.GH) ? ((unsigned)(gy * a.GW + gx0 + px) << 2) : OOB;
                    v[k] = bload(rs_p, mok ? voff : OOB, mok ? pb + ((unsigned)(m * a.GH * a.GW) << 2) : 0u);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int idx = ib + 4 * k;
                    const int m = idx >> a.lnp, rp = idx - (m << a.lnp);
                    if (idx < total) pl[((rp * 2 + prow) * WG_BW + px) * COTP + m] = v[k];
                }
            }
        }
        }
        __syncthreads();
        if (PF && tile + a.PSPLIT < ntiles) issue_tile(tile + a.PSPLIT);     // in flight during the MFMA phase
        // ---- MFMA over pixel pairs: groups of 4 steps, the next group's fragments are fetched first
        {
            float af[2][4][MT], bf[2][4][NTW];
            // one VALU add per group and operand (lane part + scalar group part); steps and m-tiles are immediates.
            // (a VALU instruction between MFMAs costs 6-8 cycles of matrix-pipe time, LDS and scalar ones none)
            const float *a_lane = pl + kh * COTP + nl;
            const float *b_lane[NTW];
#pragma unroll
            for (int u = 0; u < NTW; ++u) b_lane[u] = qp + loff[u] + kh;
            auto load = [&](auto bufc, int py, int px0) {
                constexpr int buf = decltype(bufc)::value;
                const float *ag = a_lane + (py * WG_BW + px0) * COTP;
                const int qo = (py * a.S) * a.RS + px0;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int m = 0; m < MT; ++m) af[buf][s4][m] = ag[2 * s4 * COTP + m * 32];
#pragma unroll
                for (int u = 0; u < NTW; ++u) {
                    const float *bg = b_lane[u] + qo;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) bf[buf][s4][u] = bg[2 * s4];
                }
            };
            auto mma = [&](auto bufc) {
                constexpr int buf = decltype(bufc)::value;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int u = 0; u < NTW; ++u)
                            acc[m][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][s4][m], bf[buf][s4][u], acc[m][u], 0, 0, 0);
            };
            using B0 = std::integral_constant<int, 0>;
            using B1 = std::integral_constant<int, 1>;
            const int ngroups = DVF_DBG(a, 4) ? 0 : a.BH * 4;          // 32 pixels per row = 16 steps = 4 groups of 4 steps
            load(B0{}, 0, 0);
            for (int gq = 0; gq < ngroups; gq += 2) {
                load(B1{}, (gq + 1) >> 2, ((gq + 1) & 3) * 8);
                mma(B0{});
                if (gq + 2 < ngroups) load(B0{}, (gq + 2) >> 2, ((gq + 2) & 3) * 8);
                mma(B1{});
            }
        }
    }
    // ---- add the G tile: row = P channel, col = (c, tap)
    if (DVF_DBG(a, 8)) {                                   // (keep the accumulators alive)
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[m][u][r];
        if (s == 1.2345e-30f) a.G[0] = s;
        return;
    }
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        if (gcol[u] < 0) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (mm < a.M) atomicAdd(a.G + (int64_t)(a.g_mbase + mm) * a.g_mstride + gcol[u], acc[m][u][r]);
            }
    }
}

// dpre = dY * act'(Y) and dbias[c] += sum dpre  (one pass; dbias zeroed by the caller side of the ABI)
__device__ __forceinline__ float act_grad(float g, float yv, int act, float alpha, float beta) {
    if (act == DVF_ACT_RELU) return (yv > 0.f) ? g : 0.f;
    if (act == DVF_ACT_SIGMOID_AFFINE) {
        const float sg = (yv - beta) / alpha;               // sigmoid value
        return g * alpha * sg * (1.f - sg);
    }
    return g;
}

// dpre = dy * act'(y) and dbias[c] += sum(dpre) in one pass.  VEC: 16-byte lanes (HW % 4 == 0), 4 independent loads in
// flight per thread; one atomic per block.
// part != NULL: the block's sum goes to part[c * (N * chunks) + n * chunks + chunk] instead of an atomic on dbias[c]
// (bias_finish_kernel adds them in index order: run-to-run deterministic).
template <bool VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                      float *__restrict__ dpre, float *dbias, int C, int HW, int act,
                                                      float alpha, float beta, int chunks, float *part = nullptr, int nper = 0) {
    __shared__ float red[4];
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const int c = plane % C;
    const int64_t base = (int64_t)plane * HW;
    float s = 0.f;
    if (VEC) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int HW4 = HW >> 2, per = (HW4 + chunks - 1) / chunks;
        const int beg = chunk * per, end = min(HW4, beg + per);
        const f4 *dy4 = reinterpret_cast<const f4 *>(dy + base), *y4 = reinterpret_cast<const f4 *>(y ? y + base : dy + base);
        f4 *o4 = reinterpret_cast<f4 *>(dpre ? dpre + base : nullptr);
        for (int i0 = beg + threadIdx.x; i0 < end; i0 += 1024) {
            f4 g[4], yv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 256 * k;
                if (i < end) { g[k] = dy4[i]; yv[k] = (act != DVF_ACT_NONE) ? y4[i] : g[k]; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 256 * k;
                if (i < end) {
                    f4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { r[e] = act_grad(g[k][e], yv[k][e], act, alpha, beta); s += r[e]; }
                    if (dpre) o4[i] = r;
                }
            }
        }
    } else {
        const int per = (HW + chunks - 1) / chunks;
        const int beg = chunk * per, end = min(HW, beg + per);
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            const float g = act_grad(dy[base + i], (act != DVF_ACT_NONE) ? y[base + i] : 0.f, act, alpha, beta);
            if (dpre) dpre[base + i] = g;
            s += g;
        }
    }
    if (dbias) {
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = (red[0] + red[1]) + (red[2] + red[3]);
            if (part) part[(int64_t)c * nper + (plane / C) * chunks + chunk] = t;
            else atomicAdd(&dbias[c], t);
        }
    }
}

// dbias[c] += part[c][0 .. n) summed in a fixed order (64 strided lane sums, then a fixed tree); one block per channel
__global__ __launch_bounds__(64) void bias_finish_kernel(const float *__restrict__ part, float *dbias, int n) {
    const float *p = part + (int64_t)blockIdx.x * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += p[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) dbias[blockIdx.x] += s;
}

// ---------------------------------------------------------------------------------------- host planning
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

struct TilePlan { int lsw, lsh, TGX, TGY, NT; };

// pick the 32-pixel tile shape and the arrangement of a block's tiles that waste the fewest MFMA columns
TilePlan plan_tiles(int OHc, int OWc, int NT, int maxBW = 64) {
    TilePlan best{5, 0, 1, 4 * NT, NT};
    double best_cost = 1e30;
    for (int lsw = 5; lsw >= 0; --lsw) {
        const int lsh = 5 - lsw, SW = 1 << lsw, SH = 1 << lsh;
        for (int TGX = 1; TGX <= 4 * NT; TGX *= 2) {
            const int TGY = 4 * NT / TGX, BW = TGX * SW, BH = TGY * SH;
            if (BW > maxBW) continue;
            const double cov = (double)cdiv(OWc, BW) * BW * cdiv(OHc, BH) * BH;
            // small preference for wide tiles (coalesced stores) and compact patches
            const double cost = cov * (1.0 + 0.02 * lsh) + 1e-3 * (BW + BH);
            if (cost < best_cost) { best_cost = cost; best = TilePlan{lsw, lsh, TGX, TGY, NT}; }
        }
    }
    return best;
}

struct ClassSpec { int OS, py, px, IS, by, bx, TA, TB, OHc, OWc; int tapmap[49]; };

struct GatherPlan { int MT, NT, CKH; size_t lds; dim3 grid; };

// Fill the tiling fields of `a` for the classes of one op (they share one launch).  Returns 0 or an error.
int plan_gather(GatherArgs &a, const ClassSpec *cls, int ncls, GatherPlan &pl) {
    if (ncls < 1 || ncls > 4) return DVF_ERR_INVALID_ARG;
    a.ncls = ncls; a.OS = cls[0].OS; a.IS = cls[0].IS;
    int Tmax = 0, TAmax = 0, TBmax = 0, OHc = 0, OWc = 0, tap0 = 0;
    for (int i = 0; i < ncls; ++i) {
        const ClassSpec &c = cls[i];
        const int T = c.TA * c.TB;
        if (T < 1 || c.OHc <= 0 || c.OWc <= 0 || tap0 + T > 52) return DVF_ERR_INVALID_ARG;
        a.cls[i] = ClassDev{c.py, c.px, c.by, c.bx, c.TA, c.TB, c.OHc, c.OWc, tap0};
        for (int t = 0; t < T; ++t) a.tapmap[tap0 + t] = c.tapmap[t];
        tap0 += T;
        Tmax = T > Tmax ? T : Tmax; TAmax = c.TA > TAmax ? c.TA : TAmax; TBmax = c.TB > TBmax ? c.TB : TBmax;
        OHc = c.OHc > OHc ? c.OHc : OHc; OWc = c.OWc > OWc ? c.OWc : OWc;
    }
    // Tile choice: measured on MI355X, large block tiles + split-K beat small tiles without it (the per-chunk
    // staging cost is per block, so 4x smaller tiles stage 4x more per MFMA).
    const int maxBW = ((64 - 1) * cls[0].IS + TBmax > 128) ? 32 : 64;
    const int MT = a.M > 32 ? 2 : 1;
    int NT = 2;
    TilePlan tp = plan_tiles(OHc, OWc, NT, maxBW);
    int64_t nblk = (int64_t)cdiv(OWc, tp.TGX << tp.lsw) * cdiv(OHc, tp.TGY << tp.lsh) * cdiv(a.M, 32 * MT) * a.N * ncls;
    if (nblk < 768) {
        NT = 1;
        tp = plan_tiles(OHc, OWc, NT, maxBW);
        nblk = (int64_t)cdiv(OWc, tp.TGX << tp.lsw) * cdiv(OHc, tp.TGY << tp.lsh) * cdiv(a.M, 32 * MT) * a.N * ncls;
    }
    const int mtiles = cdiv(a.M, 32 * MT);
    a.lsw = tp.lsw; a.lsh = tp.lsh; a.TGX = tp.TGX;
    a.BW = tp.TGX << tp.lsw; a.BH = tp.TGY << tp.lsh;
    if ((a.BW - 1) * a.IS + TBmax > 128) return DVF_ERR_UNSUPPORTED;   // patch rows are staged in <= 2 passes of 64
    a.tilesX = cdiv(OWc, a.BW);
    a.tilesY = cdiv(OHc, a.BH);
    const int PH = (a.BH - 1) * a.IS + TAmax, PW = (a.BW - 1) * a.IS + TBmax, PWH = (PW + 1) / 2;
    a.PSmax = PH * ((a.IS == 2) ? 2 * PWH : PW);
    a.Tmax = Tmax;
    const int COTP = 32 * MT + 1;
    int maxc = 0;
    for (int s = 0; s < a.nseg; ++s) maxc = a.segC[s] > maxc ? a.segC[s] : maxc;
    // largest chunk (2,4,8,16 channels) whose LDS footprint stays <= 40 KiB (3+ blocks per CU)
    auto lds_bytes = [&](int CK) {
        const int WD = (a.w_mode == 0 ? CK : 32 * MT) * a.KK;
        return ((size_t)CK * a.PSmax + (size_t)Tmax * CK * COTP + WD + Tmax + 64) * 4;
    };
    int CK = 16;
    static const size_t g_lds_cap = (dvf_tune("DVF_G_LDS_KB") ? atoi(dvf_tune("DVF_G_LDS_KB")) : 40) * 1024;     // tuning knob
    while (CK > 2 && lds_bytes(CK) > g_lds_cap) CK >>= 1;
    while (CK > 2 && CK / 2 >= maxc) CK >>= 1;
    if (lds_bytes(CK) > 64 * 1024) return DVF_ERR_UNSUPPORTED;
    a.WD = (a.w_mode == 0 ? CK : 32 * MT) * a.KK;
    a.NCH = 0;
    for (int s = 0; s < a.nseg; ++s) a.NCH += cdiv(a.segC[s], CK);
    int KS = 1;
    if (nblk < 384) KS = (int)((512 + nblk - 1) / nblk);
    if (KS > a.NCH) KS = a.NCH;
    if (KS < 1) KS = 1;
    if ((int64_t)a.N * KS * ncls > 65535) return DVF_ERR_UNSUPPORTED;
    a.KS = KS;
    if (const char *e = dvf_tune("DVF_DBG")) a.dbg = atoi(e);     // ablation switches for tools/conv_bench.py only
    // the kernels address each tensor with 32-bit byte offsets
    for (int s2 = 0; s2 < a.nseg; ++s2)
        if ((int64_t)a.N * a.segC[s2] * a.IH * a.IW * 4 >= ((int64_t)1 << 31) - 16) return DVF_ERR_UNSUPPORTED;
    if ((int64_t)a.Mtot * a.Rtot * a.KK * 4 >= ((int64_t)1 << 31) - 16) return DVF_ERR_UNSUPPORTED;
    pl.MT = MT; pl.NT = NT; pl.CKH = CK / 2; pl.lds = lds_bytes(CK);
    pl.grid = dim3(a.tilesX * a.tilesY, mtiles, a.N * KS * ncls);
    return DVF_OK;
}

template <int MT, int NT>
int launch_gather_ck(const GatherArgs &a, const GatherPlan &pl, hipStream_t st) {
    switch (pl.CKH) {
        case 1: conv_gather_kernel<MT, NT, 1><<<pl.grid, 256, pl.lds, st>>>(a); break;
        case 2: conv_gather_kernel<MT, NT, 2><<<pl.grid, 256, pl.lds, st>>>(a); break;
        case 4: conv_gather_kernel<MT, NT, 4><<<pl.grid, 256, pl.lds, st>>>(a); break;
        case 8: conv_gather_kernel<MT, NT, 8><<<pl.grid, 256, pl.lds, st>>>(a); break;
        default: return DVF_ERR_UNSUPPORTED;
    }
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

// Run the classes of one op in ONE launch.  With split-K (or classes that do not cover every output pixel) the
// blocks accumulate with atomics into a zeroed output and bias + activation are applied by one finishing pass;
// otherwise they are fused in the epilogue.
int run_classes(const GatherArgs &base, const ClassSpec *cls, int ncls, bool covers_all, hipStream_t st,
                float *ws = nullptr, int64_t ws_floats = 0, int64_t *ws_need = nullptr);

// splitk_reduce_kernel is defined in conv_pipe_host.h (included below)
__global__ void splitk_reduce_kernel(const float *ws, float *out, const float *bias, int KS, int64_t slice, int C, int64_t HW,
                                     int act, float alpha, float beta);

int run_classes(const GatherArgs &base, const ClassSpec *cls, int ncls, bool covers_all, hipStream_t st, float *ws,
                int64_t ws_floats, int64_t *ws_need) {
    GatherArgs a = base;
    GatherPlan pl;
    int rc = plan_gather(a, cls, ncls, pl);
    if (rc) return rc;
    const bool split = !covers_all || a.KS > 1;
    const int64_t HW = (int64_t)a.OH * a.OW, total = (int64_t)a.N * a.M * HW;
    // split-K over a covering class set: with a workspace the K-splits store plain partial tiles that one pass reduces in
    // fixed order (+bias, activation) -- deterministic; without it they accumulate with float atomics into a zeroed output
    const int64_t need = (covers_all && a.KS > 1) ? (int64_t)a.KS * total : 0;
    if (ws_need) { *ws_need = need; return DVF_OK; }       // size query only
    const bool use_ws = need > 0 && ws && ws_floats >= need;
    float *out = a.out;
    if (use_ws) { a.out = ws; a.ws_slice = total; a.atomic_out = 2; }
    else {
        if (split && hipMemsetAsync(a.out, 0, sizeof(float) * total, st) != hipSuccess) return DVF_ERR_LAUNCH;
        a.atomic_out = split ? 1 : 0;
    }
    if (a.atomic_out != 0) a.mask = nullptr;               // (the caller runs the mask pass over the finished sums)
    else if (a.mask && a.mask_done) *a.mask_done = 1;
    if (pl.MT == 2 && pl.NT == 2) rc = launch_gather_ck<2, 2>(a, pl, st);
    else if (pl.MT == 2) rc = launch_gather_ck<2, 1>(a, pl, st);
    else if (pl.NT == 2) rc = launch_gather_ck<1, 2>(a, pl, st);
    else rc = launch_gather_ck<1, 1>(a, pl, st);
    if (rc) return rc;
    dvf_plan_note(DVF_K_GATHER, pl.MT, pl.NT, 1, 2 * pl.CKH, 0, a.KS, 1, 1, 256, (int)pl.lds, a.atomic_out | (a.ncls << 4));
    const int64_t nb = (total + 255) / 256;
    if (use_ws) {
        splitk_reduce_kernel<<<(int)(nb > 4096 ? 4096 : nb), 256, 0, st>>>(ws, out, a.bias, a.KS, total, a.M, HW, a.act, a.alpha,
                                                                          a.beta);
        DVF_LAUNCH_CHECK();
    } else if (split && (a.bias || a.act != DVF_ACT_NONE)) {
        bias_act_kernel<<<(int)(nb > 2048 ? 2048 : nb), 256, 0, st>>>(out, a.bias, a.M, HW, total, a.act, a.alpha,
                                                                     a.beta);
        DVF_LAUNCH_CHECK();
    }
    return DVF_OK;
}

// Taps of output-parity class `c` for the relation  Y = o*S - pad + a  <=>  o = (Y + pad - a) / S.
// Taps are listed so that o = (Y - c)/S + base + index.  Returns the tap count.
int class_taps(int c, int S, int pad, int K, int *taps, int *base) {
    int amax = -1;
    for (int a = K - 1; a >= 0; --a)
        if ((c + pad - a) % S == 0) { amax = a; break; }
    if (amax < 0) { *base = 0; return 0; }
    int cnt = 0;
    for (int a = amax; a >= 0; a -= S) taps[cnt++] = a;
    *base = (c + pad - amax) / S;     // exact division (possibly negative)
    return cnt;
}

// classes of the scatter relation for an output of size OHxOW
int scatter_classes(int S, int pad, int KH, int KW, int OH, int OW, ClassSpec *cls, bool *covers_all) {
    int n = 0;
    *covers_all = true;
    for (int cy = 0; cy < S; ++cy)
        for (int cx = 0; cx < S; ++cx) {
            int ta[7], tb[7], by, bx;
            const int na = class_taps(cy, S, pad, KH, ta, &by), nb = class_taps(cx, S, pad, KW, tb, &bx);
            const int OHc = (OH - cy + S - 1) / S, OWc = (OW - cx + S - 1) / S;
            if (OHc <= 0 || OWc <= 0) continue;
            if (na == 0 || nb == 0) { *covers_all = false; continue; }
            ClassSpec &c = cls[n++];
            c = ClassSpec{S, cy, cx, 1, by, bx, na, nb, OHc, OWc, {0}};
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < nb; ++j) c.tapmap[i * nb + j] = ta[i] * KW + tb[j];
        }
    // heaviest class first (longest-processing-time order: a launch dispatches its classes in this order, class-major, so the
    // 4-tap blocks of a 3x3 stride-2 launch start first and the 1-tap blocks fill the tail -- conv_pipe.h, blockIdx.z)
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && cls[j].TA * cls[j].TB > cls[j - 1].TA * cls[j - 1].TB; --j) {
            const ClassSpec t = cls[j]; cls[j] = cls[j - 1]; cls[j - 1] = t;
        }
    return n;
}

#include "conv_pipe_host.h"

int check_desc(const dvf_conv_desc *d) {
    if (!d) return DVF_ERR_INVALID_ARG;
    if (d->N <= 0 || d->C_in <= 0 || d->C_out <= 0 || d->H_in <= 0 || d->W_in <= 0 || d->H_out <= 0 || d->W_out <= 0)
        return DVF_ERR_INVALID_ARG;
    if (d->KH < 1 || d->KW < 1 || d->KH > 7 || d->KW > 7 || (d->stride != 1 && d->stride != 2) || d->pad < 0)
        return DVF_ERR_INVALID_ARG;
    return DVF_OK;
}

int check_segs(const dvf_conv_desc *d, const int *seg_channels, int nseg) {
    if (nseg < 1 || nseg > DVF_MAX_SEGS || !seg_channels) return DVF_ERR_INVALID_ARG;
    int tot = 0;
    for (int s = 0; s < nseg; ++s) {
        if (seg_channels[s] <= 0) return DVF_ERR_INVALID_ARG;
        tot += seg_channels[s];
    }
    return tot == d->C_in ? DVF_OK : DVF_ERR_INVALID_ARG;
}


// ---- op builders shared by the packed entry points
int make_fwd_op(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg, const float *bias,
                float *out, PipeOp &op) {
    PipeArgs &a = op.a;
    a = PipeArgs{};
    for (int s = 0; s < nseg; ++s) {
        a.in[s] = in_segs ? in_segs[s] : nullptr;
        a.segC[s] = seg_channels[s];
    }
    a.nseg = nseg; a.M = d->C_out; a.bias = bias; a.out = out;
    a.N = d->N; a.IH = d->H_in; a.IW = d->W_in; a.OH = d->H_out; a.OW = d->W_out;
    a.act = d->act; a.alpha = d->alpha; a.beta = d->beta;
    op.KK = d->KH * d->KW; op.m_base = 0; op.Mtot = d->C_out; op.Rtot = d->C_in;
    op.head_fwd = !dvf_head_fwd_applicable(d, nseg) && dvf_head_applicable(d, nseg);
    if (!d->transposed) {
        op.w_mode = 0; op.ncls = 1; op.covers = true;
        op.cls[0] = ClassSpec{1, 0, 0, d->stride, -d->pad, -d->pad, d->KH, d->KW, d->H_out, d->W_out, {0}};
        for (int t = 0; t < op.KK; ++t) op.cls[0].tapmap[t] = t;
    } else {
        op.w_mode = 1;
        op.ncls = scatter_classes(d->stride, d->pad, d->KH, d->KW, d->H_out, d->W_out, op.cls, &op.covers);
    }
    return op.ncls >= 1 ? DVF_OK : DVF_ERR_UNSUPPORTED;
}

int make_dgrad_op(const dvf_conv_desc *d, const float *dpre, float *din, int off, int segc, PipeOp &op) {
    PipeArgs &a = op.a;
    a = PipeArgs{};
    a.in[0] = dpre; a.segC[0] = d->C_out; a.nseg = 1;
    a.M = segc; a.bias = nullptr; a.out = din;
    a.N = d->N; a.IH = d->H_out; a.IW = d->W_out; a.OH = d->H_in; a.OW = d->W_in;
    a.act = DVF_ACT_NONE;
    op.KK = d->KH * d->KW; op.m_base = off; op.Mtot = d->C_in; op.Rtot = d->C_out;
    op.head_fwd = false;
    if (!d->transposed) {
        op.w_mode = 1;
        op.ncls = scatter_classes(d->stride, d->pad, d->KH, d->KW, d->H_in, d->W_in, op.cls, &op.covers);
    } else {
        op.w_mode = 0; op.ncls = 1; op.covers = true;
        op.cls[0] = ClassSpec{1, 0, 0, d->stride, -d->pad, -d->pad, d->KH, d->KW, d->H_in, d->W_in, {0}};
        for (int t = 0; t < op.KK; ++t) op.cls[0].tapmap[t] = t;
    }
    return op.ncls >= 1 ? DVF_OK : DVF_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// Forward through the unpacked kernels.  ws_need != nullptr: only report the split-K workspace the call would use.
int fwd_unpacked(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg, const float *w,
                 const float *bias, float *out, float *ws, int64_t ws_floats, int64_t *ws_need, void *stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    if (!ws_need && (!in_segs || !w || !out)) return DVF_ERR_INVALID_ARG;
    if (ws_need) *ws_need = 0;
    if (dvf_head_applicable(d, nseg)) {      // (every head: this is the unpacked route -- the wide ones normally arrive packed)
        if (ws_need) return DVF_OK;
        if (!in_segs[0]) return DVF_ERR_INVALID_ARG;
        return dvf_head_fwd(d, in_segs[0], w, bias, out, dvf_stream(stream));
    }
    if (dvf_head_wide_applicable(d, nseg)) {
        if (ws_need) return DVF_OK;
        for (int s = 0; s < nseg; ++s)
            if (!in_segs[s]) return DVF_ERR_INVALID_ARG;
        return dvf_head_fwd_segs(d, in_segs, seg_channels, nseg, w, bias, out, dvf_stream(stream));
    }
    if (dvf_dconvt_applicable(d, nseg)) {                  // thin stride-2 transposed convolutions: direct kernel
        if (ws_need) return DVF_OK;
        if (!in_segs[0]) return DVF_ERR_INVALID_ARG;
        return dvf_dconvt_fwd(d, in_segs[0], w, bias, out, dvf_stream(stream));
    }
    GatherArgs a{};
    for (int s = 0; s < nseg; ++s) {
        if (!ws_need && !in_segs[s]) return DVF_ERR_INVALID_ARG;
        a.in[s] = ws_need ? nullptr : in_segs[s];
        a.segC[s] = seg_channels[s];
    }
    a.nseg = nseg; a.w = w; a.KK = d->KH * d->KW; a.m_base = 0; a.M = d->C_out; a.bias = bias; a.out = out;
    a.Mtot = d->C_out; a.Rtot = d->C_in;
    a.N = d->N; a.IH = d->H_in; a.IW = d->W_in; a.OH = d->H_out; a.OW = d->W_out;
    a.act = d->act; a.alpha = d->alpha; a.beta = d->beta;
    ClassSpec cls[4];
    if (!d->transposed) {
        // out[co][oy][ox] = sum_ci sum_ab W[co][ci][a][b] * in[ci][oy*s - p + a][ox*s - p + b]
        a.w_mode = 0;
        cls[0] = ClassSpec{1, 0, 0, d->stride, -d->pad, -d->pad, d->KH, d->KW, d->H_out, d->W_out, {0}};
        for (int t = 0; t < a.KK; ++t) cls[0].tapmap[t] = t;
        return run_classes(a, cls, 1, true, dvf_stream(stream), ws, ws_floats, ws_need);
    }
    // out[co][Y][X] = sum_ci sum_ab W[ci][co][a][b] * in[ci][i][j],  Y = i*s - p + a   (one launch per parity)
    a.w_mode = 1;
    bool covers;
    const int ncls = scatter_classes(d->stride, d->pad, d->KH, d->KW, d->H_out, d->W_out, cls, &covers);
    return run_classes(a, cls, ncls, covers, dvf_stream(stream), ws, ws_floats, ws_need);
}
}  // namespace

extern "C" {

int dvf_conv2d_fwd_ws(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                      const float *w, const float *bias, float *out, float *ws, int64_t ws_floats, void *stream) {
    dvf_plan_reset();
    return fwd_unpacked(d, in_segs, seg_channels, nseg, w, bias, out, ws, ws_floats, nullptr, stream);
}

int dvf_conv2d_fwd(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                   const float *w, const float *bias, float *out, void *stream) {
    return dvf_conv2d_fwd_ws(d, in_segs, seg_channels, nseg, w, bias, out, nullptr, 0, stream);
}

}  // extern "C"

namespace {
// dgrad of one input segment with the unpacked weights: the head kernel for narrow segments, else conv_gather_kernel
int dgrad_segment_unpacked(const dvf_conv_desc *d, const float *dpre, const float *w, float *din, int off, int segc,
                           hipStream_t st, float *ws = nullptr, int64_t ws_floats = 0, int64_t *ws_need = nullptr,
                           const float *mask = nullptr, int *mask_done = nullptr) {
    if (ws_need) *ws_need = 0;
    if (dvf_head_seg_dgrad_applicable(d, segc)) {
        if (ws_need) return DVF_OK;
        if (mask && mask_done) *mask_done = 1;
        return dvf_head_seg_dgrad(d, dpre, w, din, off, segc, st, mask);
    }
    GatherArgs a{};
    a.in[0] = dpre; a.segC[0] = d->C_out; a.nseg = 1;
    a.w = w; a.KK = d->KH * d->KW; a.m_base = off; a.M = segc; a.bias = nullptr; a.out = din;
    a.mask = mask; a.mask_done = mask_done;
    a.Mtot = d->C_in; a.Rtot = d->C_out;
    a.N = d->N; a.IH = d->H_out; a.IW = d->W_out; a.OH = d->H_in; a.OW = d->W_in;
    a.act = DVF_ACT_NONE;
    ClassSpec cls[4];
    if (!d->transposed) {
        // din[ci][Y][X] = sum_co sum_ab W[co][ci][a][b] * dpre[co][o][..],  Y = o*s - p + a
        a.w_mode = 1;
        bool covers;
        const int ncls = scatter_classes(d->stride, d->pad, d->KH, d->KW, d->H_in, d->W_in, cls, &covers);
        return run_classes(a, cls, ncls, covers, st, ws, ws_floats, ws_need);
    }
    // din[ci][i][j] = sum_co sum_ab W[ci][co][a][b] * dpre[co][i*s - p + a][j*s - p + b]
    a.w_mode = 0;
    cls[0] = ClassSpec{1, 0, 0, d->stride, -d->pad, -d->pad, d->KH, d->KW, d->H_in, d->W_in, {0}};
    for (int t = 0; t < a.KK; ++t) cls[0].tapmap[t] = t;
    return run_classes(a, cls, 1, true, st, ws, ws_floats, ws_need);
}
// Plan + launch of the pipelined weight-gradient kernel (wgrad_pipe.hip) for one segment; DVF_ERR_UNSUPPORTED when the
// geometry is outside its envelope (the register-staged conv_wgrad_kernel then takes the segment).
static int roundup(int v, int q) { return (v + q - 1) / q * q; }
// dbias != NULL (Conv2d roles only): also dbias[m] += sum of P[m]; *bias_done says whether the launch took it.
// ws / ws_floats: scratch for the deterministic flush (wgrad_pipe.h); *ws_need (optional) receives the floats it takes --
// with st == nullptr and ws_need set the call only plans (size query) and launches nothing.
int wgrad_pipe_op(const WgradArgs &o, const float *const *qsegs, const int *qsegc, int nseg, hipStream_t st, float *dbias = nullptr,
                  bool *bias_done = nullptr, float *ws = nullptr, int64_t ws_floats = 0, int64_t *ws_need = nullptr, bool plan_only = false) {
    if (bias_done) *bias_done = false;
    static const bool off = dvf_tune("DVF_WG_PIPE") && atoi(dvf_tune("DVF_WG_PIPE")) == 0;     // tuning knob
    if (off) return DVF_ERR_UNSUPPORTED;
    const int T = o.KH * o.KW;
    if ((o.S != 1 && o.S != 2) || T > 128 || o.pad < 0 || o.pad > 4) return DVF_ERR_UNSUPPORTED;
    if ((int64_t)o.N * o.PCtot * o.GH * o.GW * 4 >= ((int64_t)1 << 31) - 16 ||
        nseg < 1 || nseg > DVF_MAX_SEGS)
        return DVF_ERR_UNSUPPORTED;                 // 32-bit byte offsets inside the kernel
    WgpArgs w{};
    w.P = o.P; w.G = o.G;
    uintptr_t align = reinterpret_cast<uintptr_t>(o.P);
    int ctot = 0;
    for (int s = 0; s < nseg; ++s) {
        if (!qsegs[s] || qsegc[s] < 1) return DVF_ERR_INVALID_ARG;
        if ((int64_t)o.N * qsegc[s] * o.QH * o.QW * 4 >= ((int64_t)1 << 31) - 16) return DVF_ERR_UNSUPPORTED;
        w.Q[s] = qsegs[s]; w.segC[s] = qsegc[s];
        ctot += qsegc[s];
        align |= reinterpret_cast<uintptr_t>(qsegs[s]);
    }
    if (ctot != o.Cq) return DVF_ERR_INVALID_ARG;
    w.nseg = nseg;
    w.PCtot = o.PCtot; w.m_base = o.m_base; w.M = o.M; w.Cq = o.Cq;
    w.g_mstride = o.g_mstride; w.g_mbase = o.g_mbase; w.g_cbase = o.g_cbase; w.KK = o.KK; w.KH = o.KH; w.KW = o.KW;
    w.N = o.N; w.GH = o.GH; w.GW = o.GW; w.QH = o.QH; w.QW = o.QW; w.S = o.S; w.pad = o.pad;
    w.XA = (o.pad + 3) & ~3;
    w.RSq = roundup(w.XA + (WGP_BW - 1) * o.S + o.KW - o.pad, 4);
    w.PHq = (WGP_BH - 1) * o.S + o.KH;
    w.x4 = (o.GW % 4 == 0 && o.QW % 4 == 0 && (align & 15) == 0) ? 1 : 0;
    if (const char *e = dvf_tune("DVF_WG_X4")) w.x4 = w.x4 && atoi(e) != 0;      // tuning knob: 0 = 4-byte DMA lanes everywhere
    const int piece = w.x4 ? 256 : 64;
    w.NPIq = cdiv(w.PHq * w.RSq, piece);
    if (w.NPIq > WGP_MAXQ) return DVF_ERR_UNSUPPORTED;
    w.PSq = w.NPIq * piece + 4;
    // Block shape: TILE*MT rows x 4*NTW*TILE columns, MFMA 32x32x2 (TILE 32) or 16x16x4 (TILE 16: same rate, a quarter of
    // the tile).  Cost ~ matrix-pipe cycles of the layer = m-blocks x channel chunks x MT x NTW x (4 | 1); the 16-wide tiles
    // only when they save >= 15 % (small row / column counts: a 32-wide tile would be mostly padding) -- per MFMA they read
    // more LDS and re-stage more of P.  Channel chunks are balanced.
    static const bool no16 = dvf_tune("DVF_WG_TILE16") && atoi(dvf_tune("DVF_WG_TILE16")) == 0;      // tuning knob
    double best_cost = 1e30;
    int TILE = 0, MT = 0, NTW = 0, CK = 0;
    for (int tile = 32; tile >= 16; tile -= 16) {
        if (tile == 16 && no16) break;
        for (int mt = 2; mt >= 1; --mt) {
            if (mt == 2 && o.M <= tile) continue;                 // (a second row tile would be empty)
            const int pf = (tile * mt / 2) * WGP_PAIR;
            for (int ntw = (tile == 16 ? 4 : 2); ntw >= 1; --ntw) {
                int ckmax = (4 * ntw * tile) / T;
                if (ckmax > o.Cq) ckmax = o.Cq;
                while (ckmax >= 1 && (size_t)2 * (pf + (size_t)(ckmax + 1) * w.PSq) * 4 > WGP_LDS_CAP) --ckmax;   // (+1: slot of ones)
                if (ckmax < 1) continue;
                const int nch = cdiv(o.Cq, ckmax);
                double cost = (double)cdiv(o.M, tile * mt) * nch * mt * ntw * (tile == 32 ? 4 : 1);
                if (tile == 16) cost *= 1.15;
                if (cost < best_cost) { best_cost = cost; TILE = tile; MT = mt; NTW = ntw; CK = cdiv(o.Cq, nch); }
            }
        }
    }
    const int MB = TILE * MT, PF = (MB / 2) * WGP_PAIR;
    if (!NTW) return DVF_ERR_UNSUPPORTED;
    w.CK = CK;
    w.mtiles = cdiv(o.M, MB);
    w.cchunks = cdiv(o.Cq, CK);
    w.tilesX = cdiv(o.GW, WGP_BW);
    w.tilesY = cdiv(o.GH, WGP_BH);
    w.ntiles = o.N * w.tilesX * w.tilesY;
    const int64_t W = (int64_t)w.mtiles * w.cchunks * w.ntiles;
    if (W >= ((int64_t)1 << 30)) return DVF_ERR_UNSUPPORTED;
    w.W = (int)W;
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    const int nblocks = w.W < ncu ? w.W : ncu;
    w.bias_col = (dbias && CK * T < 4 * NTW * TILE) ? CK * T : -1; // a spare column of the tile multiplies by ones
    w.dbias = w.bias_col >= 0 ? dbias : nullptr;
    w.QSLOTS = CK + (w.bias_col >= 0 ? 1 : 0);
    const size_t lds = (size_t)2 * (PF + (size_t)w.QSLOTS * w.PSq) * 4;
    const int64_t need = wgrad_pipe_slots(w, nblocks) * wgrad_pipe_slot_floats(MT, NTW, TILE);
    if (ws_need) *ws_need = need;
    if (plan_only) return DVF_OK;
    w.ws = (ws && ws_floats >= need) ? ws : nullptr;
    if (const char *e = dvf_tune("DVF_WG_DBG")) w.dbg = atoi(e);
    w.stamps = nullptr;
#ifdef DVF_TUNING
    if (dvf_tune("DVF_WG_STAMPS")) {   // in-kernel cycle account of the LAST wgrad_pipe launch (tools/r3/stamps.py)
        if (!g_stamp_buf && hipMalloc(&g_stamp_buf, STAMP_MAX_BLOCKS * 64) != hipSuccess) return DVF_ERR_LAUNCH;
        if ((size_t)nblocks <= STAMP_MAX_BLOCKS) {
            if (hipMemsetAsync(g_stamp_buf, 0, (size_t)nblocks * 64, st) != hipSuccess) return DVF_ERR_LAUNCH;
            w.stamps = g_stamp_buf;
            g_stamp_blocks = nblocks;
        }
    }
#endif
    int rc = dvf_wgrad_pipe_launch(w, MT, NTW, nblocks, lds, st, TILE);
    if (rc == DVF_OK && w.ws) rc = dvf_wgrad_pipe_reduce(w, MT, NTW, nblocks, st, TILE);
    if (rc == DVF_OK) dvf_plan_note(DVF_K_WGRAD_PIPE, MT | (TILE << 8), NTW, w.x4 | (w.ws ? 2 : 0), CK, o.S, w.NPIq, nblocks, (int)lds, T, w.RSq, nseg);
    if (rc == DVF_OK && bias_done) *bias_done = w.bias_col >= 0;
    return rc;
}

}  // namespace

extern "C" {

int dvf_conv2d_dgrad_ws(const dvf_conv_desc *d, const float *dpre, const float *w, float *const *din_segs,
                        const int *seg_channels, int nseg, float *ws, int64_t ws_floats, void *stream) {
    dvf_plan_reset();
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    if (!dpre || !w || !din_segs) return DVF_ERR_INVALID_ARG;
    if (dvf_head_applicable(d, nseg)) return din_segs[0] ? dvf_head_dgrad(d, dpre, w, din_segs[0], dvf_stream(stream)) : DVF_OK;   // (unpacked route)
    int off = 0;
    for (int s = 0; s < nseg; ++s) {
        const int segc = seg_channels[s];
        if (din_segs[s]) {
            rc = dgrad_segment_unpacked(d, dpre, w, din_segs[s], off, segc, dvf_stream(stream), ws, ws_floats);
            if (rc) return rc;
        }
        off += segc;
    }
    return DVF_OK;
}

int dvf_conv2d_dgrad(const dvf_conv_desc *d, const float *dpre, const float *w, float *const *din_segs,
                     const int *seg_channels, int nseg, void *stream) {
    return dvf_conv2d_dgrad_ws(d, dpre, w, din_segs, seg_channels, nseg, nullptr, 0, stream);
}

static int wgrad_impl(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg, const float *dpre,
                      float *dw, int accumulate, float *dbias, bool *bias_done, void *stream, float *ws = nullptr,
                      int64_t ws_floats = 0, int64_t *ws_need = nullptr, bool plan_only = false);

int64_t dvf_conv2d_wgrad_ws_floats(const dvf_conv_desc *d, const int *seg_channels, int nseg) {
    int64_t need = 0;
    const float *fake[DVF_MAX_SEGS];
    for (int s = 0; s < DVF_MAX_SEGS; ++s) fake[s] = reinterpret_cast<const float *>(uintptr_t(256));   // (planning reads alignment only)
    const int rc = wgrad_impl(d, fake, seg_channels, nseg, fake[0], const_cast<float *>(fake[0]), 1, nullptr, nullptr, nullptr, nullptr, 0,
                              &need, true);
    if (rc) return rc;
    const int64_t nb = dvf_act_bwd_ws_floats(d->N, d->C_out, d->H_out * d->W_out);      // (the bias pass behind it, when one runs)
    return need > nb ? need : nb;
}

int dvf_conv2d_wgrad_det(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                         const float *dpre, float *dw, int accumulate, float *dbias, int accumulate_dbias, float *ws,
                         int64_t ws_floats, void *stream) {
    if (!d) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    if (dbias && !accumulate_dbias &&
        hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)(d->C_out > 0 ? d->C_out : 0), st) != hipSuccess)
        return DVF_ERR_LAUNCH;
    bool done = false;
    const int rc = wgrad_impl(d, in_segs, seg_channels, nseg, dpre, dw, accumulate, dbias, dbias ? &done : nullptr, stream, ws, ws_floats);
    if (rc || done || !dbias) return rc;
    return dvf_act_bwd_det(dpre, nullptr, nullptr, dbias, d->N, d->C_out, d->H_out * d->W_out, DVF_ACT_NONE, 1.f, 0.f, 1, ws, ws_floats, stream);
}

int dvf_conv2d_wgrad(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                     const float *dpre, float *dw, int accumulate, void *stream) {
    return wgrad_impl(d, in_segs, seg_channels, nseg, dpre, dw, accumulate, nullptr, nullptr, stream);
}

int dvf_conv2d_wgrad_bias(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                          const float *dpre, float *dw, int accumulate, float *dbias, int accumulate_dbias, void *stream) {
    if (!dbias || !d) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    if (!accumulate_dbias && hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)(d->C_out > 0 ? d->C_out : 0), st) != hipSuccess)
        return DVF_ERR_LAUNCH;
    bool done = false;
    const int rc = wgrad_impl(d, in_segs, seg_channels, nseg, dpre, dw, accumulate, dbias, &done, stream);
    if (rc || done) return rc;
    // the weight-gradient launch had no spare column (or ran another kernel): one pass over dpre
    return dvf_act_bwd2(dpre, nullptr, nullptr, dbias, d->N, d->C_out, d->H_out * d->W_out, DVF_ACT_NONE, 1.f, 0.f, 1, stream);
}

static int wgrad_impl(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg, const float *dpre,
                      float *dw, int accumulate, float *dbias, bool *bias_done, void *stream, float *ws, int64_t ws_floats,
                      int64_t *ws_need, bool plan_only) {
    dvf_plan_reset();
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    if (!in_segs || !dpre || !dw) return DVF_ERR_INVALID_ARG;
    hipStream_t st = dvf_stream(stream);
    if (dvf_head_applicable(d, nseg)) {
        if (plan_only) { if (ws_need) *ws_need = dvf_head_wgrad_ws_floats(d); return DVF_OK; }
        if (!in_segs[0]) return DVF_ERR_INVALID_ARG;
        return dvf_head_wgrad(d, in_segs[0], dpre, dw, accumulate, st, ws, ws_floats);
    }
    const int KK = d->KH * d->KW;
    if (!plan_only && !accumulate &&
        hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->C_in * d->C_out * KK, st) != hipSuccess)
        return DVF_ERR_LAUNCH;
    for (int s = 0; s < nseg; ++s)
        if (!in_segs[s]) return DVF_ERR_INVALID_ARG;
    if (!d->transposed) {
        // one launch over the virtual concatenation: dW[co][ci][a][b] = sum dpre[co][o] * in[ci][o*s - p + (a,b)]
        WgradArgs a{};
        a.KK = KK; a.KH = d->KH; a.KW = d->KW; a.N = d->N; a.S = d->stride; a.pad = d->pad; a.G = dw;
        a.P = dpre; a.PCtot = d->C_out; a.m_base = 0; a.M = d->C_out; a.GH = d->H_out; a.GW = d->W_out;
        a.Cq = d->C_in; a.QH = d->H_in; a.QW = d->W_in;
        a.g_mstride = (int64_t)d->C_in * KK; a.g_mbase = 0; a.g_cbase = 0;
        const int prc = wgrad_pipe_op(a, in_segs, seg_channels, nseg, st, dbias, bias_done, ws, ws_floats, ws_need, plan_only);
        if (prc != DVF_ERR_UNSUPPORTED) return prc;
    }
    if (plan_only && d->transposed) {
        // transposed layers: one launch per input segment; the scratch is reused, so the largest one counts
        int64_t worst = 0;
        for (int s = 0; s < nseg; ++s) {
            WgradArgs a{};
            a.KK = KK; a.KH = d->KH; a.KW = d->KW; a.N = d->N; a.S = d->stride; a.pad = d->pad; a.G = dw;
            a.P = in_segs[s]; a.PCtot = seg_channels[s]; a.m_base = 0; a.M = seg_channels[s]; a.GH = d->H_in; a.GW = d->W_in;
            a.Cq = d->C_out; a.QH = d->H_out; a.QW = d->W_out; a.g_mstride = (int64_t)d->C_out * KK;
            const int qc = d->C_out;
            int64_t need = 0;
            if (wgrad_pipe_op(a, &dpre, &qc, 1, st, nullptr, nullptr, nullptr, 0, &need, true) == DVF_OK && need > worst) worst = need;
        }
        if (ws_need) *ws_need = worst;
        return DVF_OK;
    }
    if (plan_only) return DVF_OK;
    int off = 0;
    for (int s = 0; s < nseg; ++s) {
        const int segc = seg_channels[s];
        WgradArgs a{};
        a.KK = KK; a.KH = d->KH; a.KW = d->KW; a.N = d->N; a.S = d->stride; a.pad = d->pad; a.G = dw;
        if (!d->transposed) {
            // dW[co][ci][a][b] = sum dpre[co][o] * in[ci][o*s - p + (a,b)]
            a.P = dpre; a.PCtot = d->C_out; a.m_base = 0; a.M = d->C_out; a.GH = d->H_out; a.GW = d->W_out;
            a.Q = in_segs[s]; a.QCtot = segc; a.q_base = 0; a.Cq = segc; a.QH = d->H_in; a.QW = d->W_in;
            a.g_mstride = (int64_t)d->C_in * KK; a.g_mbase = 0; a.g_cbase = off;
        } else {
            // dW[ci][co][a][b] = sum in[ci][i] * dpre[co][i*s - p + (a,b)]
            a.P = in_segs[s]; a.PCtot = segc; a.m_base = 0; a.M = segc; a.GH = d->H_in; a.GW = d->W_in;
            a.Q = dpre; a.QCtot = d->C_out; a.q_base = 0; a.Cq = d->C_out; a.QH = d->H_out; a.QW = d->W_out;
            a.g_mstride = (int64_t)d->C_out * KK; a.g_mbase = off; a.g_cbase = 0;
            const int qc = d->C_out;
            const int prc = wgrad_pipe_op(a, &dpre, &qc, 1, st, nullptr, nullptr, ws, ws_floats);
            if (prc == DVF_OK) { off += segc; continue; }
            if (prc != DVF_ERR_UNSUPPORTED) return prc;
        }
        const int MT = a.M > 32 ? 2 : 1;
        // column tiles per wave: two when the kernel is large, or when that lets fewer channel chunks (each of which
        // re-reads the whole P tensor) cover the segment
        static const int ntw_mode = dvf_tune("DVF_WG_NTW2") ? atoi(dvf_tune("DVF_WG_NTW2")) : 0;     // tuning knob
        int NTW = KK >= 25 ? 2 : 1;
        if (NTW == 1 && KK <= 128) {
            const int n1 = cdiv(a.Cq, 128 / KK), n2 = cdiv(a.Cq, 256 / KK);
            if ((ntw_mode == 1 && n2 == 1 && n1 > 1) || (ntw_mode == 2 && n2 < n1) || (ntw_mode == 3 && n2 < n1 && a.Cq * KK <= 1024))
                NTW = 2;
        }
        int CK = (128 * NTW) / KK;
        if (CK < 1) CK = 1;
        if (CK > a.Cq) CK = a.Cq;
        a.CK = CK;
        a.BH = a.GH <= 2 ? 2 : 4;
        a.lnp = ilog2(a.BH >> 1);
        a.tilesX = cdiv(a.GW, WG_BW);
        a.tilesY = cdiv(a.GH, a.BH);
        a.PHq = (a.BH - 1) * a.S + a.KH;
        a.PWq = (WG_BW - 1) * a.S + a.KW;
        a.PWH = (a.PWq + 1) / 2;
        a.RS = (a.S == 2) ? 2 * a.PWH : a.PWq;
        a.PS = a.PHq * a.RS;
        a.COTP = 32 * MT + 1;
        // the block's LDS (Q patch + transposed P tile) may take the whole 64 KiB a block can address with immediate
        // offsets: two blocks per CU, but as many patch channels per block as fit -- every channel chunk re-reads the
        // whole P tensor, and fewer chunks beat a third resident block (swept 36..64 KB on cfg 2: 44 KB 9.02 ms/step,
        // 52 KB 8.57, 56 KB 8.47, 64 KB 8.43)
        static const size_t wg_lds_cap = (dvf_tune("DVF_WG_LDS_KB") ? atoi(dvf_tune("DVF_WG_LDS_KB")) : 64) * 1024;
        while (CK > 1 && ((size_t)CK * a.PS + (size_t)a.BH * WG_BW * a.COTP) * 4 > wg_lds_cap) --CK;
        a.CK = CK;
        const int mtiles = cdiv(a.M, 32 * MT), cchunks = cdiv(a.Cq, CK);
        const int ntiles = a.N * a.tilesX * a.tilesY;
        static const int wg_target = dvf_tune("DVF_WG_BLOCKS") ? atoi(dvf_tune("DVF_WG_BLOCKS")) : 512;    // one round of 2 blocks per CU (swept on cfg 2: 512..2048)
        int psplit = wg_target / (mtiles * cchunks);
        if (psplit < 1) psplit = 1;
        if (psplit > ntiles) psplit = ntiles;
        if (psplit > 65535 || cchunks > 65535) return DVF_ERR_UNSUPPORTED;
        a.PSPLIT = psplit;
        const size_t lds = ((size_t)CK * a.PS + (size_t)a.BH * WG_BW * a.COTP) * 4;
        if (lds > 80 * 1024) return DVF_ERR_UNSUPPORTED;
        {   // more than the default 64 KiB of dynamic LDS needs the per-function opt-in (once per process)
            static const bool raised = [] {
                const void *fns[] = {(const void *)&conv_wgrad_kernel<2, 2, true>,  (const void *)&conv_wgrad_kernel<2, 1, true>,
                                     (const void *)&conv_wgrad_kernel<1, 2, true>,  (const void *)&conv_wgrad_kernel<1, 1, true>,
                                     (const void *)&conv_wgrad_kernel<2, 2, false>, (const void *)&conv_wgrad_kernel<2, 1, false>,
                                     (const void *)&conv_wgrad_kernel<1, 2, false>, (const void *)&conv_wgrad_kernel<1, 1, false>};
                bool ok = true;
                for (const void *f : fns)
                    ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
                return ok;
            }();
            if (!raised && lds > 64 * 1024) return DVF_ERR_UNSUPPORTED;
        }
        if ((int64_t)a.N * a.PCtot * a.GH * a.GW * 4 >= ((int64_t)1 << 31) - 16 ||
            (int64_t)a.N * a.QCtot * a.QH * a.QW * 4 >= ((int64_t)1 << 31) - 16)
            return DVF_ERR_UNSUPPORTED;             // 32-bit byte offsets inside the kernel
        const dim3 grid(mtiles, cchunks, psplit);
        // register prefetch of the next tile when a wave's share of the Q patch fits the register budget (one pass)
        static const bool no_pf = dvf_tune("DVF_WG_NOPF") != nullptr;
        const bool pf = !no_pf && a.PWq <= 64 && cdiv(CK * a.PHq, 4) <= 24 && 8 * MT * (a.BH >> 1) <= 16 * MT;
        static const bool no_vp = dvf_tune("DVF_WG_NOVP") != nullptr;
        a.vp = (pf && !no_vp && a.GW % 4 == 0 && (reinterpret_cast<uintptr_t>(a.P) & 15) == 0) ? 1 : 0;
        if (const char *e = dvf_tune("DVF_WG_DBG")) a.dbg = atoi(e);
        dvf_plan_note(DVF_K_WGRAD, MT, NTW, pf ? 1 : 0, CK, a.BH, psplit, a.vp, a.S, 256, (int)lds, KK);
        if (pf) {
            if (MT == 2 && NTW == 2) conv_wgrad_kernel<2, 2, true><<<grid, 256, lds, st>>>(a);
            else if (MT == 2) conv_wgrad_kernel<2, 1, true><<<grid, 256, lds, st>>>(a);
            else if (NTW == 2) conv_wgrad_kernel<1, 2, true><<<grid, 256, lds, st>>>(a);
            else conv_wgrad_kernel<1, 1, true><<<grid, 256, lds, st>>>(a);
        } else {
            if (MT == 2 && NTW == 2) conv_wgrad_kernel<2, 2, false><<<grid, 256, lds, st>>>(a);
            else if (MT == 2) conv_wgrad_kernel<2, 1, false><<<grid, 256, lds, st>>>(a);
            else if (NTW == 2) conv_wgrad_kernel<1, 2, false><<<grid, 256, lds, st>>>(a);
            else conv_wgrad_kernel<1, 1, false><<<grid, 256, lds, st>>>(a);
        }
        DVF_LAUNCH_CHECK();
        off += segc;
    }
    return DVF_OK;
}

static int act_bwd_chunks(int N, int C, int HW, bool has_bias) {
    int chunks = (HW + 8191) / 8192;
    const int64_t planes = (int64_t)N * C;
    while (chunks > 1 && planes * chunks > 16384) chunks >>= 1;
    // small layers: still fill the GPU -- but every block ends in ONE float atomic on dbias[c], and atomics on one address
    // serialise (~40 ns each: 1024 blocks of a 1-channel head cost 41 us for a 10 MB pass): at most ~128 blocks per channel
    while (planes * chunks < 1024 && HW / (chunks * 2) >= 1024 && (!has_bias || (int64_t)N * chunks * 2 <= 128)) chunks *= 2;
    return chunks;
}

int64_t dvf_act_bwd_ws_floats(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return DVF_ERR_INVALID_ARG;
    return (int64_t)C * N * act_bwd_chunks(N, C, HW, true);
}

int dvf_act_bwd_det(const float *dy, const float *y, float *dpre, float *dbias, int N, int C, int HW, int act, float alpha,
                    float beta, int accumulate_dbias, float *ws, int64_t ws_floats, void *stream) {
    if (!dy || (act != DVF_ACT_NONE && !y) || N <= 0 || C <= 0 || HW <= 0) return DVF_ERR_INVALID_ARG;
    if (!dbias || !ws || ws_floats < dvf_act_bwd_ws_floats(N, C, HW))
        return dvf_act_bwd2(dy, y, dpre, dbias, N, C, HW, act, alpha, beta, accumulate_dbias, stream);
    hipStream_t st = dvf_stream(stream);
    if (!accumulate_dbias && hipMemsetAsync(dbias, 0, sizeof(float) * C, st) != hipSuccess) return DVF_ERR_LAUNCH;
    const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) |
                                        reinterpret_cast<uintptr_t>(dpre)) % 16 == 0);
    const int chunks = act_bwd_chunks(N, C, HW, true);
    const int64_t planes = (int64_t)N * C;
    if (vec) act_bwd_kernel<true><<<(unsigned)(planes * chunks), 256, 0, st>>>(dy, y, dpre, dbias, C, HW, act, alpha, beta, chunks, ws, N * chunks);
    else act_bwd_kernel<false><<<(unsigned)(planes * chunks), 256, 0, st>>>(dy, y, dpre, dbias, C, HW, act, alpha, beta, chunks, ws, N * chunks);
    DVF_LAUNCH_CHECK();
    bias_finish_kernel<<<C, 64, 0, st>>>(ws, dbias, N * chunks);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_act_bwd2(const float *dy, const float *y, float *dpre, float *dbias, int N, int C, int HW, int act, float alpha,
                 float beta, int accumulate_dbias, void *stream) {
    if (!dy || (act != DVF_ACT_NONE && !y) || N <= 0 || C <= 0 || HW <= 0) return DVF_ERR_INVALID_ARG;
    if (!dpre && !dbias) return DVF_OK;
    hipStream_t st = dvf_stream(stream);
    if (dbias && !accumulate_dbias && hipMemsetAsync(dbias, 0, sizeof(float) * C, st) != hipSuccess) return DVF_ERR_LAUNCH;
    const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) |
                                        reinterpret_cast<uintptr_t>(dpre)) % 16 == 0);
    const int chunks = act_bwd_chunks(N, C, HW, dbias != nullptr);
    const int64_t planes = (int64_t)N * C;
    if (vec) act_bwd_kernel<true><<<(unsigned)(planes * chunks), 256, 0, st>>>(dy, y, dpre, dbias, C, HW, act, alpha, beta, chunks);
    else act_bwd_kernel<false><<<(unsigned)(planes * chunks), 256, 0, st>>>(dy, y, dpre, dbias, C, HW, act, alpha, beta, chunks);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_act_bwd(const float *dy, const float *y, float *dpre, float *dbias, int N, int C, int HW, int act, float alpha,
                float beta, void *stream) {
    return dvf_act_bwd2(dy, y, dpre, dbias, N, C, HW, act, alpha, beta, 0, stream);
}

}  // extern "C"

namespace {
// ---- packed-weight fast path (conv_pipe_kernel)
int64_t pipe_sizes(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind, bool want_ws) {
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    PipeOp op;
    int64_t total = 0, nf = 0, wf = 0, wmax = 0;
    int nsup = 0;
    if (op_kind == 0 && !want_ws && dvf_dconvt_applicable(d, nseg)) return DVF_ERR_UNSUPPORTED;   // (forward runs the direct kernel)
    if (op_kind == 0) {
        rc = make_fwd_op(d, nullptr, seg_channels, nseg, nullptr, nullptr, op);
        if (rc) return rc;
        rc = pipe_pack(op, nullptr, nullptr, &nf, &wf, nullptr);
        if (rc == DVF_ERR_UNSUPPORTED && want_ws) {            // unpacked kernels: their split-K scratch
            int64_t need = 0;
            rc = fwd_unpacked(d, nullptr, seg_channels, nseg, nullptr, nullptr, nullptr, nullptr, 0, &need, nullptr);
            return rc ? rc : need;
        }
        return rc ? rc : (want_ws ? wf : nf);
    }
    if (op_kind != 1) return DVF_ERR_INVALID_ARG;
    int off = 0;
    for (int s = 0; s < nseg; ++s) {
        rc = make_dgrad_op(d, nullptr, nullptr, off, seg_channels[s], op);
        if (rc) return rc;
        rc = pipe_pack(op, nullptr, nullptr, &nf, &wf, nullptr);
        if (rc && rc != DVF_ERR_UNSUPPORTED) return rc;
        if (!rc) {                                           // (unsupported segments run unpacked: no packed share)
            total += nf;
            wmax = wf > wmax ? wf : wmax;
            ++nsup;
        } else if (want_ws) {
            int64_t need = 0;
            rc = dgrad_segment_unpacked(d, nullptr, nullptr, nullptr, off, seg_channels[s], nullptr, nullptr, 0, &need);
            if (rc) return rc;
            wmax = need > wmax ? need : wmax;
        }
        off += seg_channels[s];
    }
    if (!nsup && !want_ws) return DVF_ERR_UNSUPPORTED;
    return want_ws ? wmax : total;
}

}  // namespace

extern "C" {

int64_t dvf_conv2d_packed_floats(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind) {
    return pipe_sizes(d, seg_channels, nseg, op_kind, false);
}

int64_t dvf_conv2d_ws_floats(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind) {
    return pipe_sizes(d, seg_channels, nseg, op_kind, true);
}

int dvf_conv2d_pack(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind, const float *w, float *packed,
                    void *stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    if (!w || !packed) return DVF_ERR_INVALID_ARG;
    PipeOp op;
    int64_t nf = 0;
    if (op_kind == 0) {
        rc = make_fwd_op(d, nullptr, seg_channels, nseg, nullptr, nullptr, op);
        if (rc) return rc;
        return pipe_pack(op, w, packed, &nf, nullptr, dvf_stream(stream));
    }
    if (op_kind != 1) return DVF_ERR_INVALID_ARG;
    int off = 0;
    for (int s = 0; s < nseg; ++s) {
        rc = make_dgrad_op(d, nullptr, nullptr, off, seg_channels[s], op);
        if (rc) return rc;
        rc = pipe_pack(op, w, packed, &nf, nullptr, dvf_stream(stream));
        if (rc && rc != DVF_ERR_UNSUPPORTED) return rc;
        if (!rc) packed += nf;
        off += seg_channels[s];
    }
    return DVF_OK;
}

int dvf_conv2d_fwd_packed(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                          const float *packed, const float *bias, float *out, float *ws, int64_t ws_floats, void *stream) {
    dvf_plan_reset();
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    if (!in_segs || !packed || !out) return DVF_ERR_INVALID_ARG;
    for (int s = 0; s < nseg; ++s)
        if (!in_segs[s]) return DVF_ERR_INVALID_ARG;
    PipeOp op;
    rc = make_fwd_op(d, in_segs, seg_channels, nseg, bias, out, op);
    if (rc) return rc;
    return pipe_run(op, packed, ws, ws_floats, dvf_stream(stream));
}

// dgrad of every requested segment; packed == NULL: unpacked kernels only.  mask_segs[s] != NULL: segment s was produced
// by a ReLU layer whose output is mask_segs[s]; its gradient leaves already multiplied by relu'() and its channel sums are
// added to dbias_segs[s] (fused in the pipelined kernel / its split-K finish, a pass of act_bwd_kernel after the others).
static int dgrad_all(const dvf_conv_desc *d, const float *dpre, const float *packed, const float *w, float *const *din_segs,
                     const int *seg_channels, int nseg, float *ws, int64_t ws_floats, const float *const *mask_segs,
                     float *const *dbias_segs, hipStream_t st) {
    const int HWin = d->H_in * d->W_in;
    auto postpass = [&](int s) -> int {
        if (!mask_segs || !mask_segs[s] || !din_segs[s]) return DVF_OK;
        return dvf_act_bwd2(din_segs[s], mask_segs[s], din_segs[s], dbias_segs ? dbias_segs[s] : nullptr, d->N, seg_channels[s],
                            HWin, DVF_ACT_RELU, 1.f, 0.f, 1, st);
    };
    if (dvf_head_dgrad_applicable(d, nseg) || (dvf_head_applicable(d, nseg) && !packed)) {
        if (!din_segs[0]) return DVF_OK;
        if (!w) return DVF_ERR_INVALID_ARG;
        return dvf_head_dgrad(d, dpre, w, din_segs[0], st, mask_segs ? mask_segs[0] : nullptr);     // (mask applied in the kernel)
    }
    PipeOp op;
    int off = 0;
    for (int s = 0; s < nseg; ++s) {
        int rc = DVF_ERR_UNSUPPORTED;
        int64_t nf = 0;
        if (packed) {
            rc = make_dgrad_op(d, dpre, din_segs[s], off, seg_channels[s], op);
            if (rc) return rc;
            rc = pipe_pack(op, nullptr, nullptr, &nf, nullptr, nullptr);      // plan only: is this segment packed?
            if (rc && rc != DVF_ERR_UNSUPPORTED) return rc;
        }
        auto unpacked = [&]() -> int {
            if (!din_segs[s]) return DVF_OK;
            if (!w) return DVF_ERR_INVALID_ARG;
            int done = 0;
            int r = dgrad_segment_unpacked(d, dpre, w, din_segs[s], off, seg_channels[s], st, ws, ws_floats, nullptr,
                                           mask_segs ? mask_segs[s] : nullptr, &done);
            if (r) return r;
            return done ? DVF_OK : postpass(s);
        };
        if (rc == DVF_ERR_UNSUPPORTED) {                                      // narrow segment: unpacked kernels
            rc = unpacked();
            if (rc) return rc;
        } else {
            if (din_segs[s]) {
                op.a.mask = (mask_segs && mask_segs[s]) ? mask_segs[s] : nullptr;
                op.a.dbias = (op.a.mask && dbias_segs) ? dbias_segs[s] : nullptr;
                rc = pipe_run(op, packed, ws, ws_floats, st);
                // refused at run time (an operand of a 16-byte-lane plan is not 16-byte aligned): this segment runs unpacked,
                // masked or not -- the caller must not have to know which of its segments were packed
                if (rc == DVF_ERR_UNSUPPORTED && w) rc = unpacked();
                if (rc) return rc;
            }
            packed += nf;
        }
        off += seg_channels[s];
    }
    return DVF_OK;
}

int dvf_conv2d_dgrad_packed(const dvf_conv_desc *d, const float *dpre, const float *packed, const float *w,
                            float *const *din_segs, const int *seg_channels, int nseg, float *ws, int64_t ws_floats,
                            void *stream) {
    dvf_plan_reset();
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    if (!dpre || !packed || !din_segs) return DVF_ERR_INVALID_ARG;
    if (dvf_head_dgrad_applicable(d, nseg)) return DVF_ERR_UNSUPPORTED;
    return dgrad_all(d, dpre, packed, w, din_segs, seg_channels, nseg, ws, ws_floats, nullptr, nullptr, dvf_stream(stream));
}

int dvf_conv2d_dgrad_masked(const dvf_conv_desc *d, const float *dpre, const float *packed, const float *w,
                            float *const *din_segs, const int *seg_channels, int nseg, float *ws, int64_t ws_floats,
                            const float *const *mask_segs, float *const *dbias_segs, void *stream) {
    dvf_plan_reset();
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    if (!dpre || !din_segs || !mask_segs || (!packed && !w)) return DVF_ERR_INVALID_ARG;
    return dgrad_all(d, dpre, packed, w, din_segs, seg_channels, nseg, ws, ws_floats, mask_segs, dbias_segs, dvf_stream(stream));
}


// ---- batched packing: every convolution's job in one launch per optimizer step
int dvf_conv2d_pack_jobs(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind, const float *w, float *packed,
                         void *jobs_host, int max_jobs, int *blocks_out, int *lds_bytes_out) {
    static_assert(sizeof(PackArgs) <= DVF_PACK_JOB_BYTES, "DVF_PACK_JOB_BYTES too small");
    int rc = check_desc(d);
    if (rc) return rc;
    rc = check_segs(d, seg_channels, nseg);
    if (rc) return rc;
    if (!w || !packed || !jobs_host || !blocks_out || max_jobs < 1) return DVF_ERR_INVALID_ARG;
    char *out = static_cast<char *>(jobs_host);
    PipeOp op;
    int64_t nf = 0;
    int njobs = 0;
    auto emit = [&](PipeOp &o, float *dst) -> int {
        if (njobs >= max_jobs) return DVF_ERR_INVALID_ARG;
        PackArgs job{};
        int r = pipe_pack(o, w, dst, &nf, nullptr, nullptr, &job, &blocks_out[njobs]);
        if (r) return r;
        memset(out + (size_t)njobs * DVF_PACK_JOB_BYTES, 0, DVF_PACK_JOB_BYTES);
        memcpy(out + (size_t)njobs * DVF_PACK_JOB_BYTES, &job, sizeof(job));
        if (lds_bytes_out) { const int l = 32 * ((job.CK * job.KK) | 1) * 4; if (l > *lds_bytes_out) *lds_bytes_out = l; }
        ++njobs;
        return DVF_OK;
    };
    if (op_kind == 0) {
        rc = make_fwd_op(d, nullptr, seg_channels, nseg, nullptr, nullptr, op);
        if (rc) return rc;
        rc = emit(op, packed);
        return rc ? rc : njobs;
    }
    if (op_kind != 1) return DVF_ERR_INVALID_ARG;
    int off = 0;
    for (int s = 0; s < nseg; ++s) {
        rc = make_dgrad_op(d, nullptr, nullptr, off, seg_channels[s], op);
        if (rc) return rc;
        rc = emit(op, packed);
        if (rc && rc != DVF_ERR_UNSUPPORTED) return rc;
        if (!rc) packed += nf;
        off += seg_channels[s];
    }
    return njobs;
}

int dvf_conv2d_pack_batch(const void *jobs_dev, const int *block_prefix_dev, const int *block_job_dev, int njobs,
                          int total_blocks, int lds_bytes, void *stream) {
    if (!jobs_dev || !block_prefix_dev || njobs < 1 || total_blocks < 1 || lds_bytes < 4 || lds_bytes > 64 * 1024)
        return DVF_ERR_INVALID_ARG;
    static_assert(DVF_PACK_JOB_BYTES % alignof(PackArgs) == 0, "job stride");
    static_assert(sizeof(PackArgs) == DVF_PACK_JOB_BYTES, "job stride");
    conv_pack_batch_kernel<<<total_blocks, 256, lds_bytes, dvf_stream(stream)>>>(
        static_cast<const PackArgs *>(jobs_dev), block_prefix_dev, block_job_dev, njobs);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

#ifdef DVF_TUNING
// (tuning build only, not part of include/dvf_hip.h) the cycle account of the last pipelined launch made with DVF_STAMPS
// set: 8 x u64 per block (conv_pipe.h), copied to the host after a device synchronisation.  Returns the block count.
int dvf_tuning_read_stamps(unsigned long long *out, int max_blocks) {
    if (!out || !g_stamp_buf) return 0;
    const int nb = g_stamp_blocks < max_blocks ? g_stamp_blocks : max_blocks;
    if (hipDeviceSynchronize() != hipSuccess) return DVF_ERR_LAUNCH;
    if (hipMemcpy(out, g_stamp_buf, (size_t)nb * 64, hipMemcpyDeviceToHost) != hipSuccess) return DVF_ERR_LAUNCH;
    return nb;
}
#endif

}  // extern "C"
