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

// one LDS-DMA piece (64 lanes x LB bytes); the lane width must be a literal of the builtin
template <int LB>
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t r, float *dst, unsigned voff, unsigned soff) {
    if constexpr (LB == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)dst, 16, voff, soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)dst, 4, voff, soff, 0, 0);
}
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
    // cycle account (tuning build): [0] prologue [1] item loop [2] of it: MFMA wave 0 at the barriers [3] of it: flushes
    // [4] 100 MHz ticks entry -> end [5] producer 0: waiting for its DMA [6] at the barriers [7] issuing
    unsigned long long t_entry = 0, r_entry = 0;
    DVF_STAMP(a, t_entry);
    DVF_STAMP_RT(a, r_entry);
    int mc = w0 / a.ntiles;                // output block: m-block = mc % mtiles, channel chunk = mc / mtiles
    int tile = w0 - mc * a.ntiles;

    if (wave >= 4) {
        // ================================================================== producers: all LDS-DMA loads
        // the four producers (one per SIMD) take a quarter of the pieces of every item each; two LDS stages, so a
        // producer never has more than one item in flight and "my share has landed" is a plain vmcnt(0)
        // The producers are the YOUNGER waves of their SIMDs: at equal priority every vector instruction of theirs waits for
        // a gap in the MFMA wave's stream -- ~300 cycles per v_cndmask on the cycle account (profiles/r03_wgrad_stamps_*),
        // which made a dozen offset selects per item cost as much as the item's 256 MFMAs.  Priority outranks age.
        __builtin_amdgcn_s_setprio(3);
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
        const __amdgpu_buffer_rsrc_t rs_zero_dbg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.P), 0, 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_p = DVF_DBG(a, 512) ? rs_zero_dbg : tensor_rsrc(a.P);
        const int txy = a.tilesX * a.tilesY;
        int n = tile / txy;
        int tY = (tile - n * txy) / a.tilesX;
        int tX = tile - n * txy - tY * a.tilesX;
        // Round 3 (cycle account, tools/r3/stamps.py with STAMP_KIND=wgrad, profiles/r03_wgrad_stamps_before.txt): the
        // producers were never waiting -- 15 DMA instructions per item took them as long as the item's 256 MFMAs took the
        // MFMA waves (7.6 us), and the MFMA waves stood at the barriers for 20-35 % of the loop.  Not the DMA: the
        // instruction stream around it.  Per CHANNEL the old loop re-read the segment table and the tensor pointer of the
        // virtual concatenation from the kernel-argument buffer (dependent scalar loads, lgkmcnt(0) each) and rebuilt the
        // descriptor; the piece loop tested its bound per piece.  Now the segment table lives in SGPRs (read once), a
        // descriptor is built when the channel walk ENTERS a segment, the per-channel cost is two scalar adds, and the
        // item loop is instantiated per piece count (compile-time trip counts, one contiguous stretch of code per item).
        auto U = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
        static_assert(DVF_MAX_SEGS == 5, "segment table below");
        const int sc0 = U(a.segC[0]), sc1 = U(a.nseg > 1 ? a.segC[1] : 0), sc2 = U(a.nseg > 2 ? a.segC[2] : 0),
                  sc3 = U(a.nseg > 3 ? a.segC[3] : 0), sc4 = U(a.nseg > 4 ? a.segC[4] : 0);
        const float *const q0 = a.Q[0], *const q1 = a.Q[a.nseg > 1 ? 1 : 0], *const q2 = a.Q[a.nseg > 2 ? 2 : 0],
                    *const q3 = a.Q[a.nseg > 3 ? 3 : 0], *const q4 = a.Q[a.nseg > 4 ? 4 : 0];
        auto segc_of = [&](int sg) __attribute__((always_inline)) { return sg == 0 ? sc0 : sg == 1 ? sc1 : sg == 2 ? sc2 : sg == 3 ? sc3 : sc4; };
        auto segp_of = [&](int sg) __attribute__((always_inline)) { return sg == 0 ? q0 : sg == 1 ? q1 : sg == 2 ? q2 : sg == 3 ? q3 : q4; };
        const int PSqu = U(a.PSq), NPIu = U(a.NPIq), CKu = U(a.CK), Cqu = U(a.Cq), mtu = U(a.mtiles), sfl = U(stage_floats);
        const unsigned planeQb = (unsigned)U(planeQ << 2), planePb = (unsigned)U(planeP << 2);
        unsigned long long w_p = 0, w_v = 0, w_q = 0;
        auto issue = [&](int st, auto npc, auto exactc) __attribute__((always_inline)) {
            constexpr int NP = decltype(npc)::value;
            constexpr bool EXACT = decltype(exactc)::value;
            float *Pst = smem + st * sfl;
            float *Qst = Pst + PF;
            const int mb = mc % mtu, cb = mc / mtu;
            const int m0 = mb * MB, c0 = cb * CKu;
            const int nch = min(CKu, Cqu - c0);
            const int gy0 = tY * WGP_BH, gx0 = tX * WGP_BW;
            if (DVF_DBG(a, 1)) return;
            unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            DVF_STAMP(a, s0);
            // ---- P tile: channel pairs pp = pidx, pidx+4, ...
            if (!DVF_DBG(a, 2048)) {
                const int mrem = a.M - m0;                         // valid channels of this m-block (> 0)
                const int npairs = min(MB / 2, (mrem + 1) >> 1);
                unsigned soff = ((unsigned)((n * a.PCtot + a.m_base + m0) * planeP + gy0 * a.GW + gx0) << 2) + (unsigned)(2 * pidx) * planePb;
                unsigned pvo[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    pvo[kk] = ((gy0 + p_row[kk] < a.GH) && (gx0 + p_col[kk] < a.GW)) ? p_off[kk] : OOB;
                float *dst = Pst + pidx * WGP_PAIR;
                for (int pp = pidx; pp < npairs; pp += 4) {
                    const bool odd_tail = (2 * pp + 1 >= mrem);    // second channel of the pair does not exist
                    if (a.x4) {
                        const unsigned vo = (odd_tail && lane >= 32) ? OOB : pvo[0];
                        dma_piece<16>(rs_p, dst, vo, soff);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const unsigned vo = (odd_tail && kk >= 2) ? OOB : pvo[kk];
                            dma_piece<4>(rs_p, dst + (kk << 6), vo, soff);
                        }
                    }
                    soff += 8u * planePb;
                    dst += 4 * WGP_PAIR;
                }
            }
            DVF_STAMP(a, s1);
            // ---- Q patch: channels ci = pidx, pidx+4, ...
            if (!DVF_DBG(a, 4096)) {
                const int qy0 = gy0 * S - a.pad, qx0 = gx0 * S - a.XA;
                unsigned qvo[NP];
#pragma unroll
                for (int kk = 0; kk < NP; ++kk) {
                    qvo[kk] = OOB;
                    if (EXACT || kk < NPIu) {
                        const int iy = qy0 + q_row[kk], ix = qx0 + q_col[kk];
                        const bool ok = (iy >= 0) && (iy < a.QH) && (ix >= 0) && (ix < a.QW);
                        qvo[kk] = ok ? ((unsigned)(iy * a.QW + ix) << 2) : OOB;
                    }
                }
                DVF_STAMP(a, s2);
                // virtual concatenation: walk the segments that the chunk's channels c0 .. c0+nch-1 fall into
                int ci = pidx, seg = 0, seg_first = 0;
                float *dst = Qst + pidx * PSqu;
                while (ci < nch) {
                    const int vc = c0 + ci;
                    while (vc >= seg_first + segc_of(seg)) { seg_first += segc_of(seg); ++seg; }
                    const int sgc = segc_of(seg);
                    const int ci_end = min(nch, seg_first + sgc - c0);                 // this chunk's channels inside the segment
                    const __amdgpu_buffer_rsrc_t rs_q = DVF_DBG(a, 1024) ? rs_zero_dbg : tensor_rsrc(segp_of(seg));
                    unsigned soff = (unsigned)(n * sgc + (vc - seg_first)) * planeQb;
                    for (; ci < ci_end; ci += 4) {
#pragma unroll
                        for (int kk = 0; kk < NP; ++kk)
                            if (EXACT || kk < NPIu) {
                                if (a.x4) dma_piece<16>(rs_q, dst + (kk << 8), qvo[kk], soff);
                                else dma_piece<4>(rs_q, dst + (kk << 6), qvo[kk], soff);
                            }
                        soff += 4u * planeQb;
                        dst += 4 * PSqu;
                    }
                }
                DVF_STAMP(a, s3);
                w_p += s1 - s0; w_v += s2 - s1; w_q += s3 - s2;
            }
        };
        auto advance = [&]() __attribute__((always_inline)) {
            ++tile;
            if (++tX == a.tilesX) {
                tX = 0;
                if (++tY == a.tilesY) { tY = 0; ++n; }
            }
            if (tile == a.ntiles) { tile = 0; n = 0; ++mc; }
        };
        if (DVF_DBG(a, 64)) return;        // (ablation: dispatch + prologue)
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, w_vm = 0, w_bar = 0, w_iss = 0;
        // item x lives in stage x & 1 and is issued one item ahead
        auto item_loop = [&](auto npc, auto exactc) __attribute__((always_inline)) {
            issue(0, npc, exactc);
            advance();
            for (int x = 0; x < nitems; ++x) {
                // item x must have landed before anyone passes this barrier; item x-1 is fully consumed after it, which frees
                // the stage item x+1 goes to
                DVF_STAMP(a, p0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                DVF_STAMP(a, p1);
                __builtin_amdgcn_s_barrier();
                DVF_STAMP(a, p2);
                if (x + 1 < nitems) {
                    issue((x + 1) & 1, npc, exactc);
                    advance();
                }
                DVF_STAMP(a, p3);
                w_vm += p1 - p0; w_bar += p2 - p1; w_iss += p3 - p2;
            }
        };
        using std::integral_constant;
        switch (NPIu) {
            case 1: item_loop(integral_constant<int, 1>{}, std::true_type{}); break;
            case 2: item_loop(integral_constant<int, 2>{}, std::true_type{}); break;
            case 3: item_loop(integral_constant<int, 3>{}, std::true_type{}); break;
            case 4: item_loop(integral_constant<int, 4>{}, std::true_type{}); break;
            default:
                if (NPIu <= 8) item_loop(integral_constant<int, 8>{}, std::false_type{});
                else item_loop(integral_constant<int, WGP_MAXQ>{}, std::false_type{});
        }
        if (DVF_STAMPS_ON && a.stamps && pidx == 0 && lane == 0) {
            a.stamps[8 * b + 5] = w_vm; a.stamps[8 * b + 6] = w_bar; a.stamps[8 * b + 7] = w_iss;
            if (DVF_DBG(a, 256)) { a.stamps[8 * b + 5] = w_p; a.stamps[8 * b + 6] = w_v; a.stamps[8 * b + 7] = w_q; }   // (split of the issue time: P tile | offsets | Q patch)
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

    // vy / vg: rows and pixel groups per row of the tile that lie inside the P grid.  The groups outside are zeros (the DMA's
    // range check): on the 2x7 ... 4x13 maps of the deep layers they are 50-88 % of the tile, so the K loop runs over the
    // valid groups only (an odd count is padded with one group of zeros).
    auto consume = [&](int st, int vy, int vg) {
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
        int ly = 0, ltq = 0;                                   // load iterator over the valid groups (scalar)
        auto load = [&](auto bufc) {
            constexpr int buf = decltype(bufc)::value;
            const int y = ly, tq = ltq;
            if (++ltq == vg) { ltq = 0; ++ly; }
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
        static_assert(GPR * GP == WGP_BW, "groups tile a row");
        const int NGv = vy * vg;                               // (past the last valid row the loads read rows of zeros / the next slot: unused)
        load(B0{});
        for (int it = 0; it < NGv; it += 2) {
            load(B1{});
            mma(B0{});
            if (it + 2 < NGv) load(B0{});
            mma(B1{});
        }
    };
    // add the accumulators to output block `mcv` (row = P channel, column = (c, tap)) and clear them
    auto flush = [&](int mcv) {
        if (a.ws) {
            // plain stores, 1 KiB per wave-instruction: slab[((m * NTW + u) * NACC + r) * 256 + wave * 64 + lane]
            float *slab = a.ws + (int64_t)(b + mcv) * (MB * 4 * NTW * TILE) + wave * 64 + lane;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int u = 0; u < NTW; ++u)
#pragma unroll
                    for (int r = 0; r < NACC; ++r) {
                        slab[((m * NTW + u) * NACC + r) * 256] = DVF_DBG(a, 8) ? 0.f : acc[m][u][r];
                        acc[m][u][r] = 0.f;
                    }
            return;
        }
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
    unsigned long long t_loop0 = 0, t_loop1 = 0, tb0 = 0, tb1 = 0, tf0 = 0, tf1 = 0, w_bar = 0, w_flush = 0;
    DVF_STAMP(a, t_loop0);
    int ctY, ctX;                          // tile coordinates of the current item inside its image
    {
        const int txy = a.tilesX * a.tilesY, r = tile % txy;
        ctY = r / a.tilesX;
        ctX = r - ctY * a.tilesX;
    }
    constexpr int GPX = 4 * (64 / TILE);   // pixels per MFMA group (consume())
    for (int x = 0; x < nitems; ++x) {
        const int vy = min(WGP_BH, a.GH - ctY * WGP_BH);
        const int vg = (min(WGP_BW, a.GW - ctX * WGP_BW) + GPX - 1) / GPX;
        if (++ctX == a.tilesX) { ctX = 0; if (++ctY == a.tilesY) ctY = 0; }
        // item x has landed (its producers waited for it) and item x-1 is fully consumed.  No vmcnt wait here: the
        // atomics of a flush stay in flight across the barrier.
        DVF_STAMP(a, tb0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        DVF_STAMP(a, tb1);
        w_bar += tb1 - tb0;
        if (!DVF_DBG(a, 4)) consume(x & 1, vy, vg);
        int mc_next = mc;
        if (++tile == a.ntiles) { tile = 0; ++mc_next; }
        if ((x == nitems - 1 || mc_next != mc) && !DVF_DBG(a, 128)) {
            DVF_STAMP(a, tf0);
            flush(mc);
            DVF_STAMP(a, tf1);
            w_flush += tf1 - tf0;
        }
        mc = mc_next;
    }
    DVF_STAMP(a, t_loop1);
    if (DVF_STAMPS_ON && a.stamps && wave == 0) {
        unsigned long long r_end = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DVF_STAMP_RT(a, r_end);
        if (lane == 0) {
            unsigned long long *o = a.stamps + 8 * b;
            o[0] = t_loop0 - t_entry; o[1] = t_loop1 - t_loop0; o[2] = w_bar; o[3] = w_flush; o[4] = r_end - r_entry;
        }
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

// Second half of the deterministic flush: G (+)= the slabs of every output block, summed in block order.  One thread per
// slab element e = ((m * NTW + u) * NACC + r) * 256 + wave * 64 + lane of one output block mc; the blocks whose item
// ranges [W*b/nb, W*(b+1)/nb) intersect mc's items [mc*ntiles, (mc+1)*ntiles) are consecutive.  The final add into G is
// one float atomic per element and launch (G may be shared with another launch on another stream; two contributions
// commute exactly).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgpArgs a, int MT, int NTW, int TILE, int nb) {
    const int NACC = TILE == 32 ? 16 : 4, MB = TILE * MT, slot = MB * 4 * NTW * TILE;
    const int mc = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    // blocks that hold items of mc (the same for the whole block: one thread does the 64-bit divisions):
    // first b with w1(b) > mc*ntiles ... last b with w0(b) < (mc+1)*ntiles
    __shared__ int brange[2];
    if (threadIdx.x == 0) {
        const int64_t lo = (int64_t)mc * a.ntiles, hi = lo + a.ntiles, Wt = a.W;
        int b0 = (int)((lo * nb) / Wt);
        while (b0 > 0 && (Wt * b0) / nb > lo) --b0;
        while ((Wt * (b0 + 1)) / nb <= lo) ++b0;
        int b1 = (int)((hi * nb) / Wt);
        if (b1 > nb) b1 = nb;
        while (b1 < nb && (Wt * b1) / nb < hi) ++b1;
        while (b1 > b0 + 1 && (Wt * (b1 - 1)) / nb >= hi) --b1;
        brange[0] = b0;
        brange[1] = b1;                      // exclusive
    }
    __syncthreads();
    if (e >= slot) return;
    const int lane = e & 63, wave = (e >> 6) & 3, q = e >> 8;
    const int r = q % NACC, mu = q / NACC, u = mu % NTW, m = mu / NTW;
    const int nl = lane & (TILE - 1), kh = lane / TILE;
    const int row = (TILE == 32 ? m * 32 + (r & 3) + 8 * (r >> 2) : m * 16 + r) + 4 * kh;
    const int j = (wave * NTW + u) * TILE + nl;
    const int mb = mc % a.mtiles, cb = mc / a.mtiles;
    const int m0 = mb * MB, c0 = cb * a.CK, nch = min(a.CK, a.Cq - c0);
    if (m0 + row >= a.M) return;
    const int T = a.KH * a.KW;
    const int cj = j / T, tj = j - cj * T;
    const bool is_bias = (j == a.bias_col);
    if (!is_bias && !(cj < a.CK && cj < nch)) return;
    if (is_bias && (cb != 0 || !a.dbias)) return;
    const int b0 = brange[0], b1 = brange[1];
    const float *src = a.ws + (int64_t)(b0 + mc) * slot + e;
    float s = 0.f;
    for (int b = b0; b < b1; ++b, src += slot) s += *src;
    if (is_bias) atomicAdd(a.dbias + m0 + row, s);
    else atomicAdd(a.G + (int64_t)(a.g_mbase + m0 + row) * a.g_mstride + (int64_t)(a.g_cbase + c0 + cj) * a.KK + tj, s);
}

}  // namespace

int dvf_wgrad_pipe_reduce(const WgpArgs &a, int MT, int NTW, int nblocks, hipStream_t st, int tile) {
    const int slot = tile * MT * 4 * NTW * tile;
    wgrad_reduce_kernel<<<dim3((slot + 255) / 256, a.mtiles * a.cchunks), 256, 0, st>>>(a, MT, NTW, tile, nblocks);
    return hipGetLastError() == hipSuccess ? DVF_OK : DVF_ERR_LAUNCH;
}

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
