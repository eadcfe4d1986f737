// Pipelined gather-form convolution kernel (shared by conv.hip and the conv_pipe_inst*.hip instantiation units).
//
// GEMM view: M = output channels (MFMA rows), N = 32-pixel tiles (one pixel per lane -> coalesced NCHW stores),
// K = (channel pair) x taps, v_mfma_f32_32x32x2_f32 (exact fp32).  The kernel is built around LDS-DMA
// (buffer_load ... lds): while the MFMAs consume reduction chunk g from one LDS stage, the loads of chunk g+1 land in
// the other stage without passing through registers, and their issue is interleaved with the MFMAs of chunk g so that
// even one wave per SIMD keeps the matrix pipe busy.  Two things make the DMA form possible:
//
//  * weights are PRE-PACKED (conv_pack_kernel in conv.hip) into the exact LDS image of a chunk,
//        Wp[class][m-block][chunk][kernel row][cp-group][tap in row][m-tile][lane][VW]
//    (lane = 32*(r&1) + (m&31), VW channel pairs per lane, rows zero-padded from TB to TBU taps), so a chunk's slab
//    is one contiguous run fetched 1 KiB per wave-instruction and an MFMA A-fragment for VW channel pairs is ONE
//    conflict-free ds_read_b128 / b64 at an immediate offset;
//  * the input patch is staged as "pieces" of 64 consecutive LDS floats whose per-lane source offsets (image border ->
//    out-of-range -> hardware zero fill, stride-2 de-interleave) are computed once per block and re-used for every
//    channel of every chunk; only the scalar channel base changes.
//
// The inner loop is organised in UNITS of one kernel row (TBU taps) x VW channel pairs with no branches inside a unit.
#pragma once
#include "dvf_common.h"

namespace dvfp {

// In-kernel cycle stamps (tuning build only; the product kernel contains none): s_memtime + its own lgkmcnt(0).
#ifdef DVF_TUNING
#define DVF_STAMP(a, var) do { if ((a).stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); } while (0)
#define DVF_STAMP_RT(a, var) do { if ((a).stamps) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); } while (0)
#define DVF_STAMPS_ON 1
#else
#define DVF_STAMP(a, var) do { } while (0)
#define DVF_STAMP_RT(a, var) do { } while (0)
#define DVF_STAMPS_ON 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

// Raw buffer descriptor over a whole tensor, built from wave-uniform values (readfirstlane keeps hipcc from
// wrapping every load in a waterfall loop).  num_records is 2^31-16: the per-lane offset (voffset) of a valid element is
// always below it, and an invalid element is requested at voffset 0x80000000, which the hardware range check turns
// into a load of 0.0f -- exactly the zero padding the LDS images need.  The scalar offset (soffset) carries the
// row base; it is added to the address but is not part of the range check.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tensor_rsrc(const void *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFF0, 0x00020000);
}
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ __forceinline__ float apply_act(float v, int act, float alpha, float beta) {
    if (act == DVF_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DVF_ACT_SIGMOID_AFFINE) return alpha * (1.f / (1.f + expf(-v))) + beta;
    return v;
}

struct PClass { int py, px, by, bx, TA, TB, OHc, OWc, SL, NTG; unsigned wp_off; };   // SL: slab floats; wp_off: floats

struct PipeArgs {
    const float *in[DVF_MAX_SEGS];
    int segC[DVF_MAX_SEGS];
    int nseg;
    const float *wp;         // packed weights of this launch
    const float *bias;
    float *out;              // [N, M, OH, OW]  (or the split-K workspace [KS][N, M, OH, OW] when out_mode == 2)
    int M, N, IH, IW, OH, OW;
    int OS, IS, ncls;
    PClass cls[4];
    int act;
    float alpha, beta;
    int KS, NCH, out_mode, NG;            // out_mode 0: bias+act store, 1: atomicAdd, 2: plain partial store at ks*ws_slice
    int64_t ws_slice;
    int lsw, lsh, TGX, TGY, BW, BH, BN, tilesX, tilesY;
    int x4;                               // patch DMA in 16-byte lanes (see pipe_geo)
    int blk;                              // blocked accumulation: the accumulators are banked every blk chunks (0: never)
    int SLmax, PSRmax, NST;               // LDS carve per stage (NST = 2 or 3 stages): SLmax weight floats | CK * PSRmax patch floats
    int dbg;                              // ablation switches (tools/conv_bench.py): 1 no patch loads, 2 no weight loads, 4 no MFMA
    unsigned long long *stamps;           // -DDVF_TUNING builds: per-block cycle account (8 x u64 per block), or NULL
    // dgrad of a segment produced by a ReLU layer (out_mode 0): out = (mask > 0) ? v : 0 and dbias[channel] += sum(out) --
    // the activation backward pass and the bias gradient of the PRODUCING layer, done where its gradient is born
    const float *mask;                    // same shape as out (the producing layer's output), or NULL
    float *dbias;                         // [M] accumulated with float atomics, or NULL
};

__host__ __device__ inline int pipe_row_stride(int RSu, int SW, int SH, int IS) {
    // ds_read_b32 serves lanes 0-31 in one LDS cycle when they hit 32 distinct banks: rows of the SW x SH sub-tile are
    // IS*RS floats apart, so IS*RS == SW (mod 32) is ideal.  Padding costs staging instructions (the patch is copied
    // in 64-float pieces), so it is only applied when it is cheap; a 2-way conflict on these reads is affordable.
    if (SH == 1 || SW >= 32 || (SW % IS) != 0) return RSu;
    const int mod = 32 / IS, want = (SW / IS) % mod;
    const int padded = RSu + ((want - RSu % mod) + mod) % mod;
    return (padded * 8 <= RSu * 9) ? padded : RSu;
}

struct PipeGeo { int PH, PW, PWH, RS, PSR, NPI, XA; };
// x4 = 0: 4-byte DMA lanes, pieces of 64 floats, stride-2 rows de-interleaved (even columns | odd columns).
// x4 = 1: 16-byte DMA lanes, pieces of 256 floats.  A patch row starts at the 16-byte-aligned image column at or left of
//         the first one the tile needs (XA = 0..3 floats of left margin) and is a multiple of 4 floats long, so a lane's four
//         floats never straddle a row or the image border (IW % 4 == 0, (BW * IS) % 4 == 0: the planner checks).  Rows are
//         kept in image order also for stride 2: the B fragments are then read with a lane stride of two floats (a 2-way
//         bank conflict on ds_read_b32, which the LDS has room for) instead of being de-interleaved by 4-byte lanes.
__host__ __device__ inline PipeGeo pipe_geo(int BH, int BW, int BN, int IS, int TA, int TB, int SW, int SH, int x4 = 0, int bx = 0) {
    PipeGeo g;
    g.PH = (BH - 1) * IS + TA;
    g.PW = (BW - 1) * IS + TB;
    g.PWH = (g.PW + 1) >> 1;
    g.XA = 0;
    if (x4) {
        g.XA = ((bx % 4) + 4) % 4;
        int rs = (g.XA + g.PW + 3) & ~3;
        // rows of the SW x SH sub-tile are IS*RS floats apart: IS*RS == IS*SW (mod 32) spreads them over the banks;
        // only when that costs at most 1/8 more LDS (RS stays a multiple of 4)
        if (SH > 1 && SW < 32 && (SW & 3) == 0) {
            int padded = rs;
            while (((padded * IS) & 31) != ((SW * IS) & 31) && padded < rs + 32) padded += 4;
            if (((padded * IS) & 31) == ((SW * IS) & 31) && padded * 8 <= rs * 9) rs = padded;
        }
        g.RS = rs;
        g.PSR = (BN * g.PH * g.RS + 255) & ~255;
        g.NPI = g.PSR >> 8;
        return g;
    }
    g.RS = pipe_row_stride(IS == 2 ? 2 * g.PWH : g.PW, SW, SH, IS);
    g.PSR = (BN * g.PH * g.RS + 63) & ~63;
    g.NPI = g.PSR >> 6;
    return g;
}

template <int V> struct fvec;
template <> struct fvec<1> { typedef float type; };
template <> struct fvec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct fvec<4> { typedef float type __attribute__((ext_vector_type(4))); };

// MT x 32 channels per wave, NT pixel tiles per wave, WM x (4/WM) MFMA waves, CK = 2*CKH channels per chunk, TBU
// taps per unit.  The block has SIX waves: waves 0-3 only read fragments from LDS and issue MFMAs; waves 4 and 5 are
// producers that issue the LDS-DMA loads of the even / odd chunks (they need no accumulators, so they can hold the
// per-lane source offset of every patch piece in registers) -- their scalar-heavy bookkeeping issues from their own
// instruction streams instead of stalling an MFMA wave's.  Two or three LDS stages: chunk g+NST-1 is issued
// while chunk g is consumed, and because each producer only ever has ONE chunk in flight, "my chunk has landed" is a
// plain vmcnt(0).
constexpr int PIPE_THREADS = 384;        // 4 MFMA waves + 2 producers; PIPE_THREADS_4P: + 4 producers (one per SIMD)
constexpr int PIPE_THREADS_4P = 512;
constexpr int PIPE_MAXNPI = 32;    // patch pieces per channel (per-channel patch <= 2048 floats)

// Blocked accumulation (PipeArgs::blk = n > 0): every n chunks the accumulators are added into a second register set and
// cleared (instantiations with BLK = 1), so a long reduction (K = channels x taps of 1152 and more) is a sum of partial sums over ~150-300 terms like a
// blocked CPU GEMM's instead of one sequential fmaf chain (a 1161-term chain is 2.6x further from fp64 than torch's CPU
// kernels, tools/r3/layer_noise.py).  The second set costs no occupancy (the kernel's register count is set by the
// producer waves' offset tables, 141-155 VGPRs in every variant) but 1-2 % of a step when every instantiation carries it.
template <int MT, int NT, int WM, int CKH, int TBU, int BLK = 0>
__global__ __launch_bounds__(PIPE_THREADS_4P) void conv_pipe_kernel(const PipeArgs a) {
    constexpr int CK = 2 * CKH, VW = CKH >= 4 ? 4 : CKH, CPG = CKH / VW, MTW = MT * WM;
    constexpr int UNITF = TBU * MTW * 64 * VW;             // packed floats per unit
    typedef typename fvec<VW>::type avec;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int stage_floats = a.SLmax + CK * a.PSRmax;
    const int tid = threadIdx.x, lane = tid & 63, nl = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int zper = gridDim.z / a.ncls;                   // class-major: every block of the heaviest class is dispatched first
    const int zc = blockIdx.z / zper, zr = blockIdx.z - zc * zper;
    const int ng = zr / a.KS, ks = zr - ng * a.KS;
    const PClass c = a.cls[zc];
    const int tX = blockIdx.x % a.tilesX, tY = blockIdx.x / a.tilesX;
    const int oy0 = tY * a.BH, ox0 = tX * a.BW, n0 = ng * a.BN;
    if (oy0 >= c.OHc || ox0 >= c.OWc) return;              // tile outside this (smaller) class: whole block exits
    const int mb = blockIdx.y;
    const int SW = 1 << a.lsw, SH = 1 << a.lsh;
    const PipeGeo geo = pipe_geo(a.BH, a.BW, a.BN, a.IS, c.TA, TBU, SW, SH, a.x4, c.bx);   // TBU >= TB columns staged
    const int RS = geo.RS, PSR = geo.PSR, PWH = geo.PWH;
    const int g_begin = (int)(((int64_t)a.NCH * ks) / a.KS), g_end = (int)(((int64_t)a.NCH * (ks + 1)) / a.KS);
    // cycle account of this block (tuning build): [0] prologue [1] chunk loop [2] of it: MFMA wave 0 at the chunk barriers
    // [3] epilogue [4] 100 MHz ticks entry -> end  [5] producer 0: waiting for its DMA (vmcnt)  [6] at the barriers  [7] issuing
    unsigned long long t_entry = 0, r_entry = 0;
    const int lin_block = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    DVF_STAMP(a, t_entry);
    DVF_STAMP_RT(a, r_entry);

    if (wave >= 4) {
        // ================================================================== producer waves: all LDS-DMA loads
        // Issuing an LDS-DMA instruction stalls the SIMD's issue port for ~40 cycles, which the MFMA wave sharing that
        // SIMD loses.  Blocks that own a whole CU anyway are launched with FOUR producers, one per SIMD: producers 2*par
        // and 2*par+1 share the chunks of parity `par`, half the pieces each.
        //
        // Round 3: the in-kernel cycle account (tools/r3/stamps.py, profiles/r03_pipe_stamps.txt) showed the MFMA waves
        // waiting at the chunk barriers for 20-60 % of the loop -- not for memory (the producers' vmcnt waits were < 1 us
        // per launch) but for the producers' own INSTRUCTION STREAM: per chunk hipcc had hoisted 32 v_mad_u64 + 32
        // v_cndmask + 30 v_readlane (the per-piece "image index x image stride" of the BN > 1 form, with its 64 lane-mask
        // SGPRs spilled) in front of a compare-and-branch chain per piece -- 0.4-0.65 us per CHANNEL whatever the lane
        // width, against 40 cycles per DMA instruction in isolation (tools/micro/dma_rate.hip).  Now every per-lane
        // quantity is formed when the producer ENTERS A SEGMENT of the (virtual) input concatenation, the piece loops have
        // compile-time trip counts, channel tails are zero-filled through a descriptor with no records (same
        // instructions, no per-lane select), and a chunk costs: 1 scalar add + 1 DMA per piece.
        // Everything the issue loops touch is wave-uniform; hipcc does not always see that (the class record is read
        // with a run-time index), and ONE value it takes for divergent turns the DMA loops into waterfall loops with
        // their addresses in VGPRs (3x slower kernels).  readfirstlane pins the scalars.
        auto U = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
        // (the producers are the younger waves of their SIMDs: at equal priority each of their vector instructions waits for
        // a gap in the MFMA wave's stream, ~300 cycles apiece on the cycle account; priority outranks age)
        __builtin_amdgcn_s_setprio(3);
        const int nprod = U((int)(blockDim.x >> 6) - 4), pidx = wave - 4;
        const int par = U(nprod == 2 ? pidx : (pidx >> 1));   // this producer owns chunks x with x % 2 == par ...
        const int half = U(nprod == 2 ? 0 : (pidx & 1)), nhalf = U(nprod == 2 ? 1 : 2);     // ... and this share of their pieces
        const int NWP = U(c.SL >> 8);                      // 1 KiB weight pieces per chunk
        const int iy0 = U(oy0 * a.IS + c.by), ix0 = U(ox0 * a.IS + c.bx);
        const int sfl = U(stage_floats), PSRu = U(PSR), NPIu = U(geo.NPI), SLu = U(c.SL), SLmaxu = U(a.SLmax);
        const unsigned wbase = (unsigned)U((int)(c.wp_off + (unsigned)(mb * a.NCH) * (unsigned)c.SL));
        const __amdgpu_buffer_rsrc_t rs_w = tensor_rsrc(a.wp);
        auto issue_w = [&](int g, int st) __attribute__((always_inline)) {
            if (DVF_DBG(a, 2)) return;
            float *wl = smem + st * sfl;
            unsigned so = ((wbase + (unsigned)g * (unsigned)SLu) << 2) + ((unsigned)half << 10);
            float *dst = wl + (half << 8);
            for (int p = half; p < NWP; p += nhalf) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t *)dst, 16, (unsigned)lane << 4, so, 0, 0);
                so += (unsigned)nhalf << 10;
                dst += nhalf << 8;
            }
        };
        // Chunk x (counted from g_begin) belongs to producer x & 1 and lives in stage x % NST; it is issued NST-1 chunks
        // ahead.  A producer never has more than one chunk in flight, so "my chunk has landed" is a plain vmcnt(0).
        const int D = U(a.NST - 1), nchunks = U(g_end - g_begin), g0 = U(g_begin);
        auto stage_of = [&](int x) __attribute__((always_inline)) { return D == 1 ? (x & 1) : x % 3; };
        if DVF_DBG(a, 8) return;
        // the weights of the first chunks do not need the per-lane patch offsets: they go out before those are computed
        for (int x = 0; x < D; ++x)
            if (x < nchunks && (x & 1) == par) issue_w(g0 + x, stage_of(x));

        // per-lane source offset of every patch piece: offset inside one image plane + (image in the batch group) x (image
        // stride of the segment the producer is in); bit 31 set (OOB) = outside the image / the batch -> reads 0.0f
        unsigned pvf[PIPE_MAXNPI];
        unsigned pbn4[PIPE_MAXNPI / 4];                    // image inside the block's batch group (8-bit fields)
#pragma unroll
        for (int q = 0; q < PIPE_MAXNPI / 4; ++q) pbn4[q] = 0u;
        {
            const int per_img = geo.PH * RS;
            const float inv_img = 1.0f / (float)per_img, inv_rs = 1.0f / (float)RS;
#pragma unroll
            for (int kk = 0; kk < PIPE_MAXNPI; ++kk) {
                pvf[kk] = OOB;
                if (kk < NPIu) {
                    // first float of this lane in piece kk: a piece is 64 lanes x 1 float, or 64 lanes x 4 floats (x4)
                    const int p = a.x4 ? (((kk << 6) + lane) << 2) : ((kk << 6) + lane);
                    // exact small-integer division through fp32 (p < 2^16): estimate, then one correction step
                    int bn = (int)((float)p * inv_img);
                    bn += (p - bn * per_img >= per_img) - (p - bn * per_img < 0);
                    const int rem = p - bn * per_img;
                    int row = (int)((float)rem * inv_rs);
                    row += (rem - row * RS >= RS) - (rem - row * RS < 0);
                    const int col = rem - row * RS;
                    int cs = col - geo.XA;                 // x4: image order, XA floats of aligned left margin
                    bool ok = (bn < a.BN) && (n0 + bn < a.N);
                    if (a.IS == 2 && !a.x4) {
                        cs = col < PWH ? 2 * col : 2 * (col - PWH) + 1;
                        ok = ok && (col < 2 * PWH);
                    }
                    if (!a.x4) ok = ok && (cs < geo.PW);
                    const int iy = iy0 + row, ix = ix0 + cs;
                    ok = ok && (iy >= 0) && (iy < a.IH) && (ix >= 0) && (ix < a.IW);
                    pvf[kk] = ok ? (unsigned)((iy * a.IW + ix) << 2) : OOB;
                    pbn4[kk >> 2] |= (ok ? (unsigned)bn : 0u) << ((kk & 3) * 8);
                }
            }
        }
        // a descriptor with zero records: every lane of a load through it is out of range and writes 0.0f -- the zero
        // fill of a segment's channel tail, with the instruction stream of a real channel
        const __amdgpu_buffer_rsrc_t rs_zero = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.wp), 0, 0, 0x00020000);
        const unsigned plane = (unsigned)U((a.IH * a.IW) << 2);
        // state of the segment the producer is in
        int iseg = -1, iseg_first = 0, iseg_end = 0, segc = 0;       // chunks [iseg_first, iseg_end) belong to segment iseg
        __amdgpu_buffer_rsrc_t rs_in = rs_zero;
        unsigned img_cur = 0u;                             // image stride (bytes) already inside pvf
        auto enter_segment = [&](int g) __attribute__((always_inline)) {
            while (g >= iseg_end) {
                ++iseg;
                iseg_first = iseg_end;
                iseg_end += (a.segC[iseg] + CK - 1) / CK;
            }
            iseg = U(iseg); iseg_first = U(iseg_first); iseg_end = U(iseg_end);
            segc = U(a.segC[iseg]);
            rs_in = tensor_rsrc(a.in[iseg]);
            const unsigned img = (unsigned)segc * plane;
            if (a.BN > 1) {                                // (valid offsets stay below 2^31: bit 31 keeps marking OOB lanes)
                const unsigned delta = img - img_cur;
#pragma unroll
                for (int kk = 0; kk < PIPE_MAXNPI; ++kk)
                    if (kk < NPIu) pvf[kk] = (pvf[kk] >> 31) ? OOB : pvf[kk] + ((pbn4[kk >> 2] >> ((kk & 3) * 8)) & 255u) * delta;
            }
            img_cur = img;
        };
        // patch pieces of chunk g: NP = compile-time piece count (exact, or a bucket bound with a run-time guard)
        auto issue_p = [&](int g, int st, auto lanec, auto npc, auto exactc) __attribute__((always_inline)) {
            constexpr int LB = decltype(lanec)::value, NP = decltype(npc)::value;
            constexpr bool EXACT = decltype(exactc)::value;
            constexpr int PSH = LB == 16 ? 8 : 6;          // log2 floats per piece
            if (g >= iseg_end) enter_segment(g);
            const int c0 = U((g - iseg_first) * CK);
            const int nch = U(min(CK, segc - c0));
            float *dst = smem + st * sfl + SLmaxu + half * PSRu;
            unsigned soff = (unsigned)U((int)((unsigned)(n0 * segc + c0 + half) * plane));
            auto dma = [&](__amdgpu_buffer_rsrc_t r, float *d, unsigned vo, unsigned so) __attribute__((always_inline)) {
                if constexpr (LB == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)d, 16, vo, so, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)d, 4, vo, so, 0, 0);
            };
            int ci = half;
            if (DVF_DBG(a, 32)) {                          // (ablation: same instructions, a linear L2-hot source)
                for (; ci < nch; ci += nhalf) {
#pragma unroll
                    for (int kk = 0; kk < NP; ++kk)
                        if (EXACT || kk < NPIu) dma(rs_in, dst + (kk << PSH), (unsigned)lane * LB, (unsigned)(kk << (PSH + 2)));
                    dst += nhalf * PSRu;
                }
            }
            for (; ci < nch; ci += nhalf) {
#pragma unroll
                for (int kk = 0; kk < NP; ++kk)
                    if (EXACT || kk < NPIu) dma(rs_in, dst + (kk << PSH), pvf[kk], soff);
                dst += nhalf * PSRu;
                soff += (unsigned)nhalf * plane;
            }
            for (; ci < CK; ci += nhalf) {                 // channel tail of the segment: zeros
#pragma unroll
                for (int kk = 0; kk < NP; ++kk)
                    if (EXACT || kk < NPIu) dma(rs_zero, dst + (kk << PSH), pvf[kk], 0u);
                dst += nhalf * PSRu;
            }
        };
        using std::integral_constant;
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, w_vm = 0, w_bar = 0, w_iss = 0;
        // The chunk loop is instantiated per (lane width, piece count): the dispatch runs ONCE per block, and what a producer
        // executes per chunk is one short contiguous stretch of code (a per-chunk switch over the variants cost 2.6 us per
        // call on the cycle account -- far-apart code, instruction fetches -- for four DMA instructions).
        auto chunk_loop = [&](auto lanec, auto npc, auto exactc) __attribute__((always_inline)) {
            if (!DVF_DBG(a, 1))
                for (int x = 0; x < D; ++x)
                    if (x < nchunks && (x & 1) == par) issue_p(g0 + x, stage_of(x), lanec, npc, exactc);
            for (int x = 0; x < nchunks; ++x) {
                // chunk x must have landed before anyone passes this barrier; chunk x-1 is fully consumed after it, which
                // frees the stage chunk x+D goes to
                DVF_STAMP(a, p0);
                if ((x & 1) == par) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                DVF_STAMP(a, p1);
                __builtin_amdgcn_s_barrier();
                DVF_STAMP(a, p2);
                const int nx = x + D;
                if (nx < nchunks && (nx & 1) == par) {
                    issue_w(g0 + nx, stage_of(nx));
                    if (!DVF_DBG(a, 1)) issue_p(g0 + nx, stage_of(nx), lanec, npc, exactc);
                }
                DVF_STAMP(a, p3);
                w_vm += p1 - p0; w_bar += p2 - p1; w_iss += p3 - p2;
            }
        };
        auto by_np = [&](auto lanec) __attribute__((always_inline)) {
            switch (NPIu) {
                case 1: chunk_loop(lanec, integral_constant<int, 1>{}, std::true_type{}); break;
                case 2: chunk_loop(lanec, integral_constant<int, 2>{}, std::true_type{}); break;
                case 3: chunk_loop(lanec, integral_constant<int, 3>{}, std::true_type{}); break;
                case 4: chunk_loop(lanec, integral_constant<int, 4>{}, std::true_type{}); break;
                case 5: chunk_loop(lanec, integral_constant<int, 5>{}, std::true_type{}); break;
                case 6: chunk_loop(lanec, integral_constant<int, 6>{}, std::true_type{}); break;
                default:
                    if (NPIu <= 12) chunk_loop(lanec, integral_constant<int, 12>{}, std::false_type{});
                    else chunk_loop(lanec, integral_constant<int, PIPE_MAXNPI>{}, std::false_type{});
            }
        };
        if (a.x4) by_np(integral_constant<int, 16>{});
        else by_np(integral_constant<int, 4>{});
        if (DVF_STAMPS_ON && a.stamps && pidx == 0 && lane == 0) {
            a.stamps[8 * lin_block + 5] = w_vm; a.stamps[8 * lin_block + 6] = w_bar; a.stamps[8 * lin_block + 7] = w_iss;
        }
        return;
    }

    // ====================================================================== MFMA waves
    const int wm = wave % WM, wn = wave / WM;
    // bias of the block's channels -> LDS (visible after the first chunk barrier): the epilogue must not chain 32
    // dependent global loads
    float *bias_lds = smem + a.NST * stage_floats;
    if (tid < 32 * MTW) {
        const int mm = mb * (32 * MTW) + tid;
        bias_lds[tid] = (a.bias && a.out_mode == 0 && mm < a.M) ? a.bias[mm] : 0.f;
    }
    const int NU = c.NTG * CPG;                            // units per chunk
    int opy[NT], opx[NT], opn[NT];
    int bj[NT][VW];                                        // B-fragment base (floats) per tile and channel pair
    const int pxl = nl & (SW - 1), pyl = (nl >> a.lsw) & (SH - 1), pnl = nl >> (a.lsw + a.lsh);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int q = wn * NT + i, tx = q % a.TGX, tyn = q / a.TGX, ty = tyn % a.TGY, tn = tyn / a.TGY;
        const int SN = 32 >> (a.lsw + a.lsh);
        opy[i] = ty * SH + pyl;
        opx[i] = tx * SW + pxl;
        opn[i] = tn * SN + pnl;
        // x4: rows in image order behind XA floats of margin (stride-2 tiles read every other float)
        const int bbase = kh * PSR + (opn[i] * geo.PH + opy[i] * a.IS) * RS + (a.x4 ? geo.XA + opx[i] * a.IS : opx[i]);
#pragma unroll
        for (int j = 0; j < VW; ++j) bj[i][j] = bbase + j * 2 * PSR;
    }
    f32x16 acc[MT][NT];
    f32x16 asum[BLK ? MT : 1][BLK ? NT : 1];              // (BLK) running sum of the banked partial sums
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][i][r] = 0.f;
                if constexpr (BLK) asum[m][i][r] = 0.f;
            }

    // The chunk in stage `st`, unit by unit.  A unit is one kernel ROW (ta) x VW channel pairs: its TBU taps are
    // consecutive floats of a patch row, so their B fragments share ONE row address per (tile, channel pair) and differ
    // by immediate offsets -- a plain VALU instruction between two MFMAs costs 6-8 cycles of matrix-pipe time on gfx950
    // (tools/micro/mfma_mix.hip), scalar and LDS instructions cost none, so address arithmetic is kept scalar except
    // for that one add per row.  The next unit's fragments are fetched behind the current unit's MFMAs
    // (unconditionally: past the end of the chunk they read LDS that nobody uses).
    // TBE = taps of this class's rows that are actually multiplied: TBU, or TBU-1 for the narrower parity classes of a strided
    // launch (their packed rows are zero-padded to TBU taps: a 3x3 stride-2 launch would spend 12 tap slots on 9 taps)
    auto consume = [&](int st, auto isc, auto tbec) {
        constexpr int IS = decltype(isc)::value, TBE = decltype(tbec)::value;
        const float *wl = smem + st * stage_floats;
        const float *patch = wl + a.SLmax;
        const float *ap = wl + (wm * MT) * 64 * VW + lane * VW;
        avec af[2][MT];                                    // A fragments: ping-pong by tap (one tap ahead)
        float bf[2][TBE][NT][VW];                          // B fragments: ping-pong by unit (one unit ahead)
        int lta = 0, lcpg = 0, lk = 0;                     // load iterator (scalar): next unit to fetch
        // B rows of the next unit for channel pair j (all tiles): TBU consecutive floats of one patch row each
        auto load_bj = [&](auto bufc, auto jc) {
            constexpr int buf = decltype(bufc)::value, j = decltype(jc)::value;
            const float *prow0 = patch + (lta * RS + lcpg * (2 * VW) * PSR);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float *prow = prow0 + bj[i][j];
                if constexpr (IS == 1) {
#pragma unroll
                    for (int u = 0; u < TBE; ++u) bf[buf][u][i][j] = prow[u];
                } else {
                    const float *prow2 = prow + PWH;       // odd columns of the de-interleaved row
#pragma unroll
                    for (int u = 0; u < TBE; ++u) bf[buf][u][i][j] = (u & 1) ? prow2[u >> 1] : prow[u >> 1];
                }
            }
        };
        auto load_b = [&](auto bufc) {
            load_bj(bufc, std::integral_constant<int, 0>{});
            if constexpr (VW > 1) load_bj(bufc, std::integral_constant<int, 1>{});
            if constexpr (VW > 2) load_bj(bufc, std::integral_constant<int, 2>{});
            if constexpr (VW > 3) load_bj(bufc, std::integral_constant<int, 3>{});
        };
        // A fragments of tap u of unit `unit` (scalar) into ping-pong buffer tp
        auto load_a = [&](auto tpc, int unit, auto uc) {
            constexpr int tp = decltype(tpc)::value, u = decltype(uc)::value;
            const float *aptr = ap + unit * UNITF;
#pragma unroll
            for (int m = 0; m < MT; ++m) af[tp][m] = *reinterpret_cast<const avec *>(aptr + (u * MTW + m) * 64 * VW);
        };
        auto advance = [&]() {
            ++lk;
            const bool wrap = (lcpg + 1 == CPG);
            lcpg = wrap ? 0 : lcpg + 1;
            lta += wrap ? 1 : 0;
        };
        // the MT*NT MFMAs of channel pair j of tap u
        auto mma_j = [&](auto bufc, auto tpc, auto uc, auto jc) {
            constexpr int buf = decltype(bufc)::value, tp = decltype(tpc)::value, u = decltype(uc)::value, j = decltype(jc)::value;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float av;
                if constexpr (VW == 1) av = af[tp][m]; else av = af[tp][m][j];
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[buf][u][i][j], acc[m][i], 0, 0, 0);
            }
        };
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        // Tap by tap: behind the first MFMA group (channel pair 0) of tap u go the A fragments of the NEXT tap (this unit's
        // tap u+1, or the next unit's tap 0) and, in tap 0, the B rows of the next unit follow pair by pair behind the
        // groups.  lk = index of the next unit.  Keeping A only one tap ahead (not a whole unit) saves 32-48 VGPRs.  The
        // scheduling barriers pin this order: hipcc otherwise sinks the reads next to their uses.  Measured in isolation
        // (tools/micro/pipe_loop.hip, profiles/r03_pipe_loop.txt): this loop runs at 66.5 cycles per MFMA (64 = the pipe's
        // rate) with the reads here OR behind the last group (round 2's order) -- the loop was never the limiter.
        auto step = [&](auto bufc, auto nbufc, auto pbc, auto uc) {
            constexpr int u = decltype(uc)::value, tp = (decltype(pbc)::value + u) & 1;
            using TP = std::integral_constant<int, tp>;
            mma_j(bufc, TP{}, uc, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u + 1 < TBE) load_a(std::integral_constant<int, tp ^ 1>{}, lk - 1, std::integral_constant<int, u + 1>{});
            else load_a(std::integral_constant<int, tp ^ 1>{}, lk, std::integral_constant<int, 0>{});
            if constexpr (u == 0) load_bj(nbufc, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (VW > 1) {
                mma_j(bufc, TP{}, uc, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (u == 0) { load_bj(nbufc, std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0); }
            }
            if constexpr (VW > 2) {
                mma_j(bufc, TP{}, uc, std::integral_constant<int, 2>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (u == 0) { load_bj(nbufc, std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0); }
            }
            if constexpr (VW > 3) {
                mma_j(bufc, TP{}, uc, std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (u == 0) { load_bj(nbufc, std::integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0); }
            }
        };
        auto unit = [&](auto bufc, auto nbufc, auto pbc) {
            step(bufc, nbufc, pbc, std::integral_constant<int, 0>{});
            if constexpr (TBE > 1) step(bufc, nbufc, pbc, std::integral_constant<int, 1>{});
            if constexpr (TBE > 2) step(bufc, nbufc, pbc, std::integral_constant<int, 2>{});
            if constexpr (TBE > 3) step(bufc, nbufc, pbc, std::integral_constant<int, 3>{});
            if constexpr (TBE > 4) step(bufc, nbufc, pbc, std::integral_constant<int, 4>{});
            if constexpr (TBE > 5) step(bufc, nbufc, pbc, std::integral_constant<int, 5>{});
            if constexpr (TBE > 6) step(bufc, nbufc, pbc, std::integral_constant<int, 6>{});
            advance();
        };
        load_b(B0{});
        load_a(B0{}, 0, std::integral_constant<int, 0>{});
        advance();
        __builtin_amdgcn_sched_barrier(0);
        for (int k = 0; k < NU; k += 2) {
            unit(B0{}, B1{}, std::integral_constant<int, 0>{});                 // taps 0 .. TBU-1: A parity starts at 0
            if (k + 1 < NU) unit(B1{}, B0{}, std::integral_constant<int, TBE & 1>{});
        }
    };

    unsigned long long t_loop0 = 0, t_loop1 = 0, tb0 = 0, tb1 = 0, w_bar = 0;
    DVF_STAMP(a, t_loop0);
    if (!DVF_DBG(a, 8)) {
        int st = 0, since = 0;
        for (int g = g_begin; g < g_end; ++g) {
            DVF_STAMP(a, tb0);
            __syncthreads();          // chunk g has landed (its producer waited for it) and chunk g-1 is fully consumed
            DVF_STAMP(a, tb1);
            w_bar += tb1 - tb0;
            if (!DVF_DBG(a, 4)) {
                using TBfull = std::integral_constant<int, TBU>;
                using TBless = std::integral_constant<int, (TBU > 1 ? TBU - 1 : 1)>;
                if (a.IS == 2 && !a.x4) consume(st, std::integral_constant<int, 2>{}, TBfull{});      // (de-interleaved rows)
                else if (TBU > 1 && c.TB == TBU - 1 && !DVF_DBG(a, 64)) consume(st, std::integral_constant<int, 1>{}, TBless{});
                else consume(st, std::integral_constant<int, 1>{}, TBfull{});
            }
            if constexpr (BLK) {
                if (++since == a.blk) {     // (uniform)
                    since = 0;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            asum[m][i] += acc[m][i];
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
                        }
                }
            }
            st = (st + 1 == a.NST) ? 0 : st + 1;
        }
    }
    if constexpr (BLK) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[m][i] += asum[m][i];
    }

    DVF_STAMP(a, t_loop1);
    // (tuning build) the account is written when the wave leaves the kernel: epilogue stores issued AND completed
    auto stamp_out = [&]() {
        if (DVF_STAMPS_ON && a.stamps && wave == 0) {
            unsigned long long t_end = 0, r_end = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DVF_STAMP(a, t_end);
            DVF_STAMP_RT(a, r_end);
            if (lane == 0) {
                unsigned long long *o = a.stamps + 8 * lin_block;
                o[0] = t_loop0 - t_entry; o[1] = t_loop1 - t_loop0; o[2] = w_bar; o[3] = t_end - t_loop1; o[4] = r_end - r_entry;
            }
        }
    };
    // ---- epilogue: D[row = channel][col = pixel]; lanes 0-31 / 32-63 hold channel rows +0 / +4
    if DVF_DBG(a, 16) { stamp_out(); return; }
    const int m0 = mb * (32 * MTW) + wm * (32 * MT);
    float *outp = a.out + (a.out_mode == 2 ? (int64_t)ks * a.ws_slice : 0);
    const int64_t HW = (int64_t)a.OH * a.OW;
    if (a.mask && a.out_mode == 0) {
        // ReLU backward of the producing layer + its bias gradient (no bias / activation of this op: it is a dgrad)
        float csum[MT][16];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) csum[m][r] = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int oy = oy0 + opy[i], ox = ox0 + opx[i], n = n0 + opn[i];
            const int Y = oy * a.OS + c.py, X = ox * a.OS + c.px;
            const bool pok = (oy < c.OHc) && (ox < c.OWc) && (Y < a.OH) && (X < a.OW) && (opn[i] < a.BN) && (n < a.N);
            const int64_t base = ((int64_t)n * a.M + m0 + 4 * kh) * HW + (int64_t)Y * a.OW + X;
            // all mask loads of the tile first (no branch around them: invalid elements read element 0), then the stores
            float mk[MT][16];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = m * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = pok && (m0 + 4 * kh + ml < a.M);
                    mk[m][r] = a.mask[ok ? base + (int64_t)ml * HW : 0];
                }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = m * 32 + (r & 3) + 8 * (r >> 2);
                    if (pok && m0 + 4 * kh + ml < a.M) {
                        const float v = mk[m][r] > 0.f ? acc[m][i][r] : 0.f;
                        outp[base + (int64_t)ml * HW] = v;
                        csum[m][r] += v;
                    }
                }
            }
        }
        if (a.dbias) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sv = csum[m][r];
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) sv += __shfl_xor(sv, off, 64);   // over the 32 pixels of this half-wave
                    const int mm = m0 + 4 * kh + m * 32 + (r & 3) + 8 * (r >> 2);
                    if (nl == 0 && mm < a.M) atomicAdd(a.dbias + mm, sv);
                }
        }
        stamp_out();
        return;
    }
    // The output mode, the activation and "every channel of this wave exists" are launch constants: the store loops are
    // instantiated per combination (evaluating them per element cost 30+ scalar branches and an inlined sigmoid in every
    // one of the 16*MT*NT stores of a lane -- microseconds per block), rows are reached by scalar multiples of HW.
    const bool full_m = m0 + 32 * MT <= a.M;
    auto store_all = [&](auto modec, auto actc, auto fullc) __attribute__((always_inline)) {
        constexpr int MODE = decltype(modec)::value, ACT = decltype(actc)::value;
        constexpr bool FULL = decltype(fullc)::value;
        float bv[MT][16];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) bv[m][r] = bias_lds[wm * (32 * MT) + 4 * kh + m * 32 + (r & 3) + 8 * (r >> 2)];
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int oy = oy0 + opy[i], ox = ox0 + opx[i], n = n0 + opn[i];
            const int Y = oy * a.OS + c.py, X = ox * a.OS + c.px;
            const bool pok = (oy < c.OHc) && (ox < c.OWc) && (Y < a.OH) && (X < a.OW) && (opn[i] < a.BN) && (n < a.N);
            float *pbase = outp + ((int64_t)n * a.M + m0 + 4 * kh) * HW + (int64_t)Y * a.OW + X;
            if (pok) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = m * 32 + (r & 3) + 8 * (r >> 2);
                        if (FULL || m0 + 4 * kh + ml < a.M) {
                            float *op = pbase + (int64_t)ml * HW;      // (ml * HW: scalar)
                            const float v = acc[m][i][r];
                            if constexpr (MODE == 0) {
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
    if (a.out_mode == 2) by_full(integral_constant<int, 2>{}, integral_constant<int, DVF_ACT_NONE>{});
    else if (a.out_mode == 1) by_full(integral_constant<int, 1>{}, integral_constant<int, DVF_ACT_NONE>{});
    else if (a.act == DVF_ACT_RELU) by_full(integral_constant<int, 0>{}, integral_constant<int, DVF_ACT_RELU>{});
    else if (a.act == DVF_ACT_SIGMOID_AFFINE) by_full(integral_constant<int, 0>{}, integral_constant<int, DVF_ACT_SIGMOID_AFFINE>{});
    else by_full(integral_constant<int, 0>{}, integral_constant<int, DVF_ACT_NONE>{});
    stamp_out();
}

// Launch one (MT, NT, WM) family; CKH in {2,4,8} and TBU in {2,3,4} are dispatched inside.  Defined in the
// conv_pipe_inst*.hip units (explicit specialisations), declared here for conv.hip.
template <int MT, int NT, int WM>
int launch_pipe_family(const PipeArgs &a, int CKH, int TBU, dim3 grid, size_t lds, hipStream_t st, int threads);

// blocks that use more than 64 KiB of LDS need the opt-in attribute (set once per kernel)
// (BLK = 1, the instantiation with the second accumulator set, exists for the shapes the product banks: 64-channel waves,
// 2- to 4-tap rows, 8- or 16-channel chunks; the tuning build has it for every shape -- DVF_PIPE_BLKT.  The other
// instantiations keep round 2's register allocation: carrying the second set everywhere cost 1-2 % of a step.)
template <int MT, int NT, int WM, int CKH, int TBU>
constexpr bool pipe_has_blk() {
#ifdef DVF_TUNING
    return true;
#else
    return MT == 2 && NT == 1 && WM == 1 && CKH >= 4 && TBU >= 2 && TBU <= 4;
#endif
}

template <int MT, int NT, int WM, int CKH, int TBU, int BLK>
inline int launch_pipe_blk(const PipeArgs &a, dim3 grid, size_t lds, hipStream_t st, int threads) {
    static bool big_lds = false;
    if (lds > 64 * 1024 && !big_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_pipe_kernel<MT, NT, WM, CKH, TBU, BLK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return DVF_ERR_LAUNCH;
        big_lds = true;
    }
    conv_pipe_kernel<MT, NT, WM, CKH, TBU, BLK><<<grid, threads, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? DVF_OK : DVF_ERR_LAUNCH;
}

template <int MT, int NT, int WM, int CKH, int TBU>
inline int launch_pipe_one(const PipeArgs &a, dim3 grid, size_t lds, hipStream_t st, int threads) {
    if (a.blk) {
        if constexpr (pipe_has_blk<MT, NT, WM, CKH, TBU>()) return launch_pipe_blk<MT, NT, WM, CKH, TBU, 1>(a, grid, lds, st, threads);
        else return DVF_ERR_UNSUPPORTED;
    }
    return launch_pipe_blk<MT, NT, WM, CKH, TBU, 0>(a, grid, lds, st, threads);
}

template <int MT, int NT, int WM, int CKH>
inline int launch_pipe_tbu(const PipeArgs &a, int TBU, dim3 grid, size_t lds, hipStream_t st, int threads) {
    switch (TBU) {
        case 1: return launch_pipe_one<MT, NT, WM, CKH, 1>(a, grid, lds, st, threads);
        case 2: return launch_pipe_one<MT, NT, WM, CKH, 2>(a, grid, lds, st, threads);
        case 3: return launch_pipe_one<MT, NT, WM, CKH, 3>(a, grid, lds, st, threads);
        case 4: return launch_pipe_one<MT, NT, WM, CKH, 4>(a, grid, lds, st, threads);
        case 5: if constexpr (CKH <= 4 && WM == 1) return launch_pipe_one<MT, NT, WM, CKH, 5>(a, grid, lds, st, threads); else return DVF_ERR_UNSUPPORTED;
        case 7: if constexpr (CKH <= 4 && WM == 1) return launch_pipe_one<MT, NT, WM, CKH, 7>(a, grid, lds, st, threads); else return DVF_ERR_UNSUPPORTED;
        default: return DVF_ERR_UNSUPPORTED;
    }
}

#define DVF_PIPE_FAMILY(MT, NT, WM)                                                                              \
    template <>                                                                                                  \
    int launch_pipe_family<MT, NT, WM>(const PipeArgs &a, int CKH, int TBU, dim3 grid, size_t lds, hipStream_t st, int threads) { \
        switch (CKH) {                                                                                           \
            case 2: return launch_pipe_tbu<MT, NT, WM, 2>(a, TBU, grid, lds, st, threads);                                \
            case 4: return launch_pipe_tbu<MT, NT, WM, 4>(a, TBU, grid, lds, st, threads);                                \
            case 8: return launch_pipe_tbu<MT, NT, WM, 8>(a, TBU, grid, lds, st, threads);                                \
            default: return DVF_ERR_UNSUPPORTED;                                                                 \
        }                                                                                                        \
    }

}  // namespace dvfp
