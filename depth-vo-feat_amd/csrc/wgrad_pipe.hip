// Pipelined weight-gradient kernel for gfx950: LDS-DMA producers + MFMA consumers, Stream-K work split.
//
//   G[m][c][ta][tb] += sum over (n, y, x) of  P[n][m][y][x] * Q[n][c][y*S + ta - pad][x*S + tb - pad]
//
// GEMM view: M = P channels (MFMA rows, 32*MT per block), N = (c, tap) columns (4 waves x NTW tiles of 32), K = pixels,
// v_mfma_f32_32x32x2_f32 (exact fp32).  One pipeline step = one 4 x 32 pixel tile of one image:
//
//  * P tile in LDS in its NATURAL layout, [channel pair][2 x 128 pixels] + 4 floats of bank skew per pair.  An A fragment
//    is one ds_read_b128 per lane = four consecutive pixels of channel `nl`; the half-wave kh = 1 reads the next four,
//    so MFMA step i of a group multiplies the pixel pair (8t + i, 8t + 4 + i).  No transposition anywhere.
//  * Q patch (tile + halo, zero padded by the buffer range check) in its natural row-major layout; the B fragment of
//    column (c, ta, tb) for that pixel pair is a ds_read_b32 at a per-lane base + an immediate.
//  * Waves 4-7 are producers: every byte enters LDS by buffer_load ... lds (16-byte lanes when the rows are 16-byte
//    aligned: one instruction per P channel pair / per 256 patch floats), two LDS stages, one barrier per tile.  Waves
//    0-3 only issue ds_read + MFMA.
//  * Stream-K: the (m-block, channel chunk, pixel tile) work items of the layer are numbered m-block fastest and cut into
//    gridDim.x equal contiguous ranges (one block per CU), so deep layers with few tiles per output block still fill
//    every CU evenly; a block adds its accumulators to G (float atomics) whenever its range leaves an output block.
#include "wgrad_pipe.h"
#include "conv_pipe.h"

namespace {

using dvfp::f32x16;
using dvfp::lds_void_t;
using dvfp::OOB;
using dvfp::tensor_rsrc;

constexpr int WGP_THREADS = 512;           // 4 MFMA waves + 4 producers (one per SIMD)
typedef float f4 __attribute__((ext_vector_type(4)));

// TILE = 32: v_mfma_f32_32x32x2_f32, 32*MT rows x 4*NTW*32 columns per block.  TILE = 16: v_mfma_f32_16x16x4_f32 (same
// rate per SIMD, four pixels per instruction), 16*MT rows x 4*NTW*16 columns -- for layers whose gradient has at most 16 rows
// (iconv1, the pose network's first convolution), where a 32-row tile is half padding.
template <int MT, int NTW, int S, int TILE>
__global__ __launch_bounds__(WGP_THREADS) void wgrad_pipe_kernel(const WgpArgs a) {
    constexpr int MB = TILE * MT, PF = (MB / 2) * WGP_PAIR;
    constexpr int KQ = 64 / TILE;              // pixel groups of 4 per MFMA step: lane -> (tile row / column = lane % TILE, kq = lane / TILE)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, nl = lane & (TILE - 1), kh = lane / TILE;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stage_floats = PF + a.QSLOTS * a.PSq;
    const int nb = gridDim.x, b = blockIdx.x;
    const int w0 = (int)(((int64_t)a.W * b) / nb), w1 = (int)(((int64_t)a.W * (b + 1)) / nb);
    const int nitems = w1 - w0;
    if (nitems <= 0) return;
    if (DVF_DBG(a, 32)) return;            // (ablation: dispatch only)
    int mc = w0 / a.ntiles;                // output block: m-block = mc % mtiles, channel chunk = mc / mtiles
    int tile = w0 - mc * a.ntiles;

    if (wave >= 4) {
        // ================================================================== producers: all LDS-DMA loads
        // the four producers (one per SIMD) take a quarter of the pieces of every item each; two LDS stages, so a
        // producer never has more than one item in flight and "my share has landed" is a plain vmcnt(0)
        const int pidx = wave - 4;
        const int planeP = a.GH * a.GW, planeQ = a.QH * a.QW;
        unsigned p_off[4];
        int p_row[4], p_col[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            int ch, row, col;
            if (a.x4) { ch = lane >> 5; row = (lane >> 3) & 3; col = (lane & 7) << 2; }
            else { ch = kk >> 1; row = ((kk & 1) << 1) + (lane >> 5); col = lane & 31; }
            p_off[kk] = (unsigned)(ch * planeP + row * a.GW + col) << 2;
            p_row[kk] = row;
            p_col[kk] = col;
        }
        int q_row[WGP_MAXQ], q_col[WGP_MAXQ];
        {
            const int per_row = a.x4 ? (a.RSq >> 2) : a.RSq;      // DMA lanes per patch row
#pragma unroll
            for (int kk = 0; kk < WGP_MAXQ; ++kk) {
                q_row[kk] = -(1 << 20);
                q_col[kk] = 0;
                if (kk < a.NPIq) {
                    const int L = (kk << 6) + lane;
                    const int row = L / per_row, col = L - row * per_row;
                    q_row[kk] = row < a.PHq ? row : -(1 << 20);   // lanes past the patch: out of range -> zero fill
                    q_col[kk] = a.x4 ? (col << 2) : col;
                }
            }
        }
        const __amdgpu_buffer_rsrc_t rs_p = tensor_rsrc(a.P);
        const int txy = a.tilesX * a.tilesY;
        int n = tile / txy;
        int tY = (tile - n * txy) / a.tilesX;
        int tX = tile - n * txy - tY * a.tilesX;
        auto issue = [&](int st) {
            float *Pst = smem + st * stage_floats;
            float *Qst = Pst + PF;
            const int mb = mc % a.mtiles, cb = mc / a.mtiles;
            const int m0 = mb * MB, c0 = cb * a.CK;
            const int nch = min(a.CK, a.Cq - c0);
            const int gy0 = tY * WGP_BH, gx0 = tX * WGP_BW;
            if (DVF_DBG(a, 1)) return;
            // ---- P tile: channel pairs pp = pidx, pidx+4, ...
            {
                const int mrem = a.M - m0;                         // valid channels of this m-block (> 0)
                const int npairs = min(MB / 2, (mrem + 1) >> 1);
                const unsigned soff0 = (unsigned)((n * a.PCtot + a.m_base + m0) * planeP + gy0 * a.GW + gx0) << 2;
                unsigned pvo[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    pvo[kk] = ((gy0 + p_row[kk] < a.GH) && (gx0 + p_col[kk] < a.GW)) ? p_off[kk] : OOB;
                for (int pp = pidx; pp < npairs; pp += 4) {
                    const bool odd_tail = (2 * pp + 1 >= mrem);    // second channel of the pair does not exist
                    const unsigned soff = soff0 + ((unsigned)(2 * pp * planeP) << 2);
                    float *dst = Pst + pp * WGP_PAIR;
                    if (a.x4) {
                        const unsigned vo = (odd_tail && lane >= 32) ? OOB : pvo[0];
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (lds_void_t *)dst, 16, vo, soff, 0, 0);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const unsigned vo = (odd_tail && kk >= 2) ? OOB : pvo[kk];
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (lds_void_t *)(dst + (kk << 6)), 4, vo, soff, 0, 0);
                        }
                    }
                }
            }
            // ---- Q patch: channels ci = pidx, pidx+4, ...
            {
                const int qy0 = gy0 * S - a.pad, qx0 = gx0 * S - a.XA;
                unsigned qvo[WGP_MAXQ];
#pragma unroll
                for (int kk = 0; kk < WGP_MAXQ; ++kk) {
                    qvo[kk] = OOB;
                    if (kk < a.NPIq) {
                        const int iy = qy0 + q_row[kk], ix = qx0 + q_col[kk];
                        const bool ok = (iy >= 0) && (iy < a.QH) && (ix >= 0) && (ix < a.QW);
                        qvo[kk] = ok ? ((unsigned)(iy * a.QW + ix) << 2) : OOB;
                    }
                }
                // virtual concatenation: channel c0 + ci of the chunk lives in segment `seg` at channel vc - seg_first
                int seg = 0, seg_first = 0;
                for (int ci = pidx; ci < nch; ci += 4) {
                    const int vc = c0 + ci;
                    while (vc >= seg_first + a.segC[seg]) { seg_first += a.segC[seg]; ++seg; }
                    const __amdgpu_buffer_rsrc_t rs_q = tensor_rsrc(a.Q[seg]);
                    const unsigned soff = (unsigned)((n * a.segC[seg] + (vc - seg_first)) * planeQ) << 2;
                    float *dst = Qst + ci * a.PSq;
                    if (a.x4) {
#pragma unroll
                        for (int kk = 0; kk < WGP_MAXQ; ++kk)
                            if (kk < a.NPIq)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_void_t *)(dst + (kk << 8)), 16, qvo[kk], soff, 0, 0);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < WGP_MAXQ; ++kk)
                            if (kk < a.NPIq)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_void_t *)(dst + (kk << 6)), 4, qvo[kk], soff, 0, 0);
                    }
                }
            }
        };
        auto advance = [&]() {
            ++tile;
            if (++tX == a.tilesX) {
                tX = 0;
                if (++tY == a.tilesY) { tY = 0; ++n; }
            }
            if (tile == a.ntiles) { tile = 0; n = 0; ++mc; }
        };
        if (DVF_DBG(a, 64)) return;        // (ablation: dispatch + prologue)
        // item x lives in stage x & 1 and is issued one item ahead
        issue(0);
        advance();
        for (int x = 0; x < nitems; ++x) {
            // item x must have landed before anyone passes this barrier; item x-1 is fully consumed after it, which frees
            // the stage item x+1 goes to
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (x + 1 < nitems) {
                issue((x + 1) & 1);
                advance();
            }
        }
        return;
    }

    // ====================================================================== MFMA waves
    const int T = a.KH * a.KW;
    int loff[NTW], cjv[NTW], tjv[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int j = (wave * NTW + u) * TILE + nl;
        const int cj = j / T, tj = j - cj * T, ta = tj / a.KW, tb = tj - ta * a.KW;
        const bool in_chunk = cj < a.CK;
        loff[u] = in_chunk ? cj * a.PSq + ta * a.RSq + tb + (a.XA - a.pad) : 0;
        cjv[u] = in_chunk ? cj : -1;
        tjv[u] = tj;
        if (j == a.bias_col) {             // bias gradient: this column reads the slot of ones behind the chunk's channels
            loff[u] = a.CK * a.PSq;
            cjv[u] = -2;
        }
    }
    if (a.bias_col >= 0) {                 // (visible to the MFMA waves after the first barrier; the DMA never writes it)
        for (int e = tid; e < a.PSq; e += 256) {
            smem[PF + a.CK * a.PSq + e] = 1.f;
            smem[stage_floats + PF + a.CK * a.PSq + e] = 1.f;
        }
    }
    constexpr int NACC = TILE == 32 ? 16 : 4;  // accumulator registers of one MFMA tile
    typedef float accv __attribute__((ext_vector_type(NACC)));
    accv acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[m][u][r] = 0.f;

    auto consume = [&](int st) {
        const float *Pst = smem + st * stage_floats;
        const float *Qst = Pst + PF;
        const float *ap = Pst + ((nl >> 1) * WGP_PAIR + (nl & 1) * 128 + 4 * kh);
        const float *bp[NTW];
#pragma unroll
        for (int u = 0; u < NTW; ++u) bp[u] = Qst + loff[u] + 4 * kh * S;
        f4 A[2][MT];
        float B[2][NTW][4];
        // group `it` = 4*KQ pixels of a tile row (8 with the 32-wide tiles, 16 with the 16-wide ones): lane (., kq) holds
        // pixels 4*kq .. 4*kq+3 of the group, MFMA step i multiplies component i of every kq
        constexpr int GP = 4 * KQ, GPR = WGP_BW / GP;          // pixels per group, groups per tile row
        auto load = [&](auto bufc, int it) {
            constexpr int buf = decltype(bufc)::value;
            const int y = it / GPR, tq = it - y * GPR;
            const float *ay = ap + y * 32 + GP * tq;
#pragma unroll
            for (int m = 0; m < MT; ++m) A[buf][m] = *reinterpret_cast<const f4 *>(ay + m * (TILE / 2) * WGP_PAIR);
            const int qo = (y * S) * a.RSq + GP * tq * S;
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const float *bg = bp[u] + qo;
#pragma unroll
                for (int i = 0; i < 4; ++i) B[buf][u][i] = bg[i * S];
            }
        };
        auto mma = [&](auto bufc) {
            constexpr int buf = decltype(bufc)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int u = 0; u < NTW; ++u) {
                        if constexpr (TILE == 32)
                            acc[m][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[buf][m][i], B[buf][u][i], acc[m][u], 0, 0, 0);
                        else
                            acc[m][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[buf][m][i], B[buf][u][i], acc[m][u], 0, 0, 0);
                    }
        };
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        constexpr int NG = WGP_BH * WGP_BW / GP;
        load(B0{}, 0);
        for (int it = 0; it < NG; it += 2) {
            load(B1{}, it + 1);
            mma(B0{});
            if (it + 2 < NG) load(B0{}, it + 2);
            mma(B1{});
        }
    };
    // add the accumulators to output block `mcv` (row = P channel, column = (c, tap)) and clear them
    auto flush = [&](int mcv) {
        const int mb = mcv % a.mtiles, cb = mcv / a.mtiles;
        const int m0 = mb * MB, c0 = cb * a.CK;
        const int nch = min(a.CK, a.Cq - c0);
        const bool full_m = m0 + MB <= a.M;                       // every row of the block exists: no per-element test
        // accumulator register r of tile m is row  m*32 + (r & 3) + 8*(r >> 2) + 4*kh  (32x32)  /  m*16 + r + 4*kh  (16x16)
        auto row_of = [](int m, int r) { return TILE == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) : m * 16 + r; };
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            const bool colok = cjv[u] >= 0 && cjv[u] < nch && !DVF_DBG(a, 8);
            // per-lane part of the address once; the row (m, r) adds a SCALAR multiple of the row stride
            float *gl = a.G + (int64_t)(a.g_mbase + m0 + 4 * kh) * a.g_mstride + (int64_t)(a.g_cbase + c0 + cjv[u]) * a.KK + tjv[u];
            if (colok) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < NACC; ++r) {
                        const int ml = row_of(m, r);
                        if (full_m || m0 + 4 * kh + ml < a.M) atomicAdd(gl + (int64_t)ml * a.g_mstride, acc[m][u][r]);
                    }
            }
            if (cjv[u] == -2 && cb == 0) {                         // bias column (every chunk computes it; chunk 0 delivers it)
                float *bl = a.dbias + m0 + 4 * kh;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < NACC; ++r) {
                        const int ml = row_of(m, r);
                        if (full_m || m0 + 4 * kh + ml < a.M) atomicAdd(bl + ml, acc[m][u][r]);
                    }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < NACC; ++r) acc[m][u][r] = 0.f;
        }
    };
    if (DVF_DBG(a, 64)) return;
    for (int x = 0; x < nitems; ++x) {
        // item x has landed (its producers waited for it) and item x-1 is fully consumed.  No vmcnt wait here: the
        // atomics of a flush stay in flight across the barrier.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!DVF_DBG(a, 4)) consume(x & 1);
        int mc_next = mc;
        if (++tile == a.ntiles) { tile = 0; ++mc_next; }
        if ((x == nitems - 1 || mc_next != mc) && !DVF_DBG(a, 128)) flush(mc);
        mc = mc_next;
    }
}

template <int MT, int NTW, int S, int TILE>
int launch_one(const WgpArgs &a, int nblocks, size_t lds, hipStream_t st) {
    static bool big_lds = false;
    if (lds > 64 * 1024 && !big_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_pipe_kernel<MT, NTW, S, TILE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)WGP_LDS_CAP) != hipSuccess)
            return DVF_ERR_LAUNCH;
        big_lds = true;
    }
    wgrad_pipe_kernel<MT, NTW, S, TILE><<<nblocks, WGP_THREADS, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? DVF_OK : DVF_ERR_LAUNCH;
}

template <int MT, int NTW, int TILE>
int launch_s(const WgpArgs &a, int nblocks, size_t lds, hipStream_t st) {
    if (a.S == 1) return launch_one<MT, NTW, 1, TILE>(a, nblocks, lds, st);
    if (a.S == 2) return launch_one<MT, NTW, 2, TILE>(a, nblocks, lds, st);
    return DVF_ERR_UNSUPPORTED;
}

}  // namespace

int dvf_wgrad_pipe_launch(const WgpArgs &a, int MT, int NTW, int nblocks, size_t lds, hipStream_t st, int tile) {
    if (nblocks < 1 || lds > WGP_LDS_CAP || a.NPIq > WGP_MAXQ) return DVF_ERR_UNSUPPORTED;
    if (tile == 16) {                      // 16*MT rows x 64*NTW columns per block
        if (MT == 1 && NTW == 1) return launch_s<1, 1, 16>(a, nblocks, lds, st);
        if (MT == 1 && NTW == 2) return launch_s<1, 2, 16>(a, nblocks, lds, st);
        if (MT == 1 && NTW == 3) return launch_s<1, 3, 16>(a, nblocks, lds, st);
        if (MT == 1 && NTW == 4) return launch_s<1, 4, 16>(a, nblocks, lds, st);
        if (MT == 2 && NTW == 1) return launch_s<2, 1, 16>(a, nblocks, lds, st);
        if (MT == 2 && NTW == 2) return launch_s<2, 2, 16>(a, nblocks, lds, st);
        if (MT == 2 && NTW == 3) return launch_s<2, 3, 16>(a, nblocks, lds, st);
        if (MT == 2 && NTW == 4) return launch_s<2, 4, 16>(a, nblocks, lds, st);
        return DVF_ERR_UNSUPPORTED;
    }
    if (MT == 2 && NTW == 2) return launch_s<2, 2, 32>(a, nblocks, lds, st);
    if (MT == 2 && NTW == 1) return launch_s<2, 1, 32>(a, nblocks, lds, st);
    if (MT == 1 && NTW == 2) return launch_s<1, 2, 32>(a, nblocks, lds, st);
    if (MT == 1 && NTW == 1) return launch_s<1, 1, 32>(a, nblocks, lds, st);
    return DVF_ERR_UNSUPPORTED;
}
