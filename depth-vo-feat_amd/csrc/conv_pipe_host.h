// Host side of the pipelined convolution (included by conv.hip inside its anonymous namespace): weight packing,
// split-K reduction, planning and launch.
using dvfp::PClass;
using dvfp::PipeArgs;
using dvfp::PipeGeo;
using dvfp::pipe_geo;
using dvfp::PIPE_MAXNPI;

// ------------------------------------------------------------------------------------------------ weight packing
struct alignas(8) PackArgsBody {
    const float *w;
    float *wp;
    int w_mode;              // 0: w[((m_base+m)*Rtot + r)*KK + tap]   1: w[(r*Mtot + m_base+m)*KK + tap]
    int Mtot, Rtot, KK, m_base, M;
    int nseg, segC[DVF_MAX_SEGS];
    int ncls, NCH, CK, MB, VW, TBU;
    int TB[4], TA[4];        // taps per kernel row / kernel rows of each class
    int SL[4];
    unsigned wp_off[4];
    signed char tapmap[4][52];   // class tap -> stored tap
};
// jobs of the batched launch are stored with a fixed DVF_PACK_JOB_BYTES stride
struct PackArgs : PackArgsBody { char pad[DVF_PACK_JOB_BYTES - sizeof(PackArgsBody)]; };

// One block per (chunk, 32-channel m-tile): the source weights of the tile (32 x CK x KK floats) are staged in LDS with
// coalesced reads, then every packed element of the tile -- for every class, including the zero padding of channel
// tails, short kernel rows and m >= M -- is written in destination order (coalesced 4-byte lanes over contiguous runs).
__device__ __forceinline__ void pack_block(const PackArgs &a, int g, int mt, float *S) {
    const int tid = threadIdx.x;
    const int MTW = a.MB >> 5, mb = mt / MTW, mtw = mt - mb * MTW, m0 = mt * 32;
    int seg_start = 0, gg = g, segc = a.segC[0];
    for (int s = 0; s < a.nseg; ++s) {
        const int nchs = (a.segC[s] + a.CK - 1) / a.CK;
        segc = a.segC[s];
        if (gg < nchs) break;
        gg -= nchs;
        seg_start += a.segC[s];
    }
    const int c0 = gg * a.CK, nch = min(a.CK, segc - c0), r0 = seg_start + c0;
    const int mvalid = min(32, a.M - m0);
    const int CKK = a.CK * a.KK;
    const int SR = CKK | 1;                                // odd LDS row stride: the m-strided reads below hit 32 banks
    // source -> LDS: the tile is walked as one flat index space, eight independent loads in flight per thread (a block
    // moves only a few KB, so its run time is the latency of its dependent round trips to HBM)
    constexpr int PU = 8;
    if (a.w_mode == 0) {
        const int run = nch * a.KK, total = mvalid * run;      // `run` contiguous floats per output channel
        const float *src = a.w + ((int64_t)(a.m_base + m0) * a.Rtot + r0) * a.KK;
        const int64_t mstride = (int64_t)a.Rtot * a.KK;
        for (int i0 = tid; i0 < total; i0 += 256 * PU) {
            float v[PU];
            int si[PU];
#pragma unroll
            for (int j = 0; j < PU; ++j) {
                const int i = i0 + j * 256;
                if (i < total) {
                    const int m = i / run, x = i - m * run;
                    v[j] = src[m * mstride + x];
                    si[j] = m * SR + x;
                }
            }
#pragma unroll
            for (int j = 0; j < PU; ++j)
                if (i0 + j * 256 < total) S[si[j]] = v[j];
        }
    } else {
        const int run = mvalid * a.KK, total = nch * run;      // `run` contiguous floats per reduction channel
        const float *src = a.w + ((int64_t)r0 * a.Mtot + a.m_base + m0) * a.KK;
        const int64_t rstride = (int64_t)a.Mtot * a.KK;
        for (int i0 = tid; i0 < total; i0 += 256 * PU) {
            float v[PU];
            int si[PU];
#pragma unroll
            for (int j = 0; j < PU; ++j) {
                const int i = i0 + j * 256;
                if (i < total) {
                    const int rl = i / run, x = i - rl * run, m = x / a.KK, tap = x - m * a.KK;
                    v[j] = src[rl * rstride + x];
                    si[j] = m * SR + rl * a.KK + tap;
                }
            }
#pragma unroll
            for (int j = 0; j < PU; ++j)
                if (i0 + j * 256 < total) S[si[j]] = v[j];
        }
    }
    __syncthreads();
    // destination order without divisions: a thread keeps its lane (and channel pair) and walks (unit, tap) incrementally
    const int CKH = a.CK >> 1, CPG = CKH / a.VW, LV = 64 * a.VW;       // LV in {128, 256}
    if (a.VW == 4) {
        // 16-byte stores: a thread owns one lane's four channel pairs of a (unit, tap); 4 (unit, tap) rows per pass
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int lane = tid & 63, kh = lane >> 5, m = lane & 31;
        for (int c = 0; c < a.ncls; ++c) {
            const int TB = a.TB[c], NKU = a.TA[c] * CPG * a.TBU;
            float *dst = a.wp + a.wp_off[c] + (int64_t)(mb * a.NCH + g) * a.SL[c] + (int64_t)mtw * 256 + lane * 4;
            int ku = tid >> 6, u = ku, cpg = 0, ta = 0;
            while (u >= a.TBU) { u -= a.TBU; if (++cpg == CPG) { cpg = 0; ++ta; } }
            for (; ku < NKU; ku += 4) {
                f4 v = {0.f, 0.f, 0.f, 0.f};
                if (u < TB && m < mvalid) {
                    const float *src = S + m * SR + a.tapmap[c][ta * TB + u];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rl = 2 * (cpg * 4 + j) + kh;
                        if (rl < nch) v[j] = src[rl * a.KK];
                    }
                }
                *reinterpret_cast<f4 *>(dst + (int64_t)ku * MTW * 256) = v;
                u += 4;
                while (u >= a.TBU) { u -= a.TBU; if (++cpg == CPG) { cpg = 0; ++ta; } }
            }
        }
        return;
    }
    const int x = tid & (LV - 1), kstep = 256 / LV;
    const int lane = x / a.VW, j = x - lane * a.VW, kh = lane >> 5, m = lane & 31;
    for (int c = 0; c < a.ncls; ++c) {
        const int TB = a.TB[c], NKU = a.TA[c] * CPG * a.TBU;
        float *dst = a.wp + a.wp_off[c] + (int64_t)(mb * a.NCH + g) * a.SL[c] + (int64_t)mtw * LV + x;
        int ku = tid / LV, u = ku, cpg = 0, ta = 0;       // ku = (ta*CPG + cpg)*TBU + u
        while (u >= a.TBU) { u -= a.TBU; if (++cpg == CPG) { cpg = 0; ++ta; } }
        for (; ku < NKU; ku += kstep) {
            const int rl = 2 * (cpg * a.VW + j) + kh;
            float v = 0.f;
            if (u < TB && rl < nch && m < mvalid) v = S[m * SR + rl * a.KK + a.tapmap[c][ta * TB + u]];
            dst[(int64_t)ku * MTW * LV] = v;
            u += kstep;
            while (u >= a.TBU) { u -= a.TBU; if (++cpg == CPG) { cpg = 0; ++ta; } }
        }
    }
}

__global__ __launch_bounds__(256) void conv_pack_kernel(const PackArgs a) {
    extern __shared__ float S[];                           // [32][CK*KK | 1]
    pack_block(a, blockIdx.x, blockIdx.y, S);
}

// All packing jobs of a training step in ONE launch: block b belongs to the job j with prefix[j] <= b < prefix[j+1],
// read from the per-block table when the caller uploaded one (one load instead of a chain of log2(njobs) dependent ones).
__global__ __launch_bounds__(256) void conv_pack_batch_kernel(const PackArgs *jobs, const int *prefix, const int *block_job,
                                                              int njobs) {
    extern __shared__ float S[];
    const int b = blockIdx.x;
    int lo = 0;
    if (block_job) {
        lo = block_job[b];
    } else {
        int hi = njobs;                                    // invariant: prefix[lo] <= b < prefix[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] <= b) lo = mid; else hi = mid;
        }
    }
    const PackArgs &a = jobs[lo];
    const int local = b - prefix[lo];
    pack_block(a, local % a.NCH, local / a.NCH, S);
}

// out = act(sum_k ws[k] + bias): finishes a split-K convolution whose blocks stored plain partial tiles.
__global__ void splitk_reduce_kernel(const float *ws, float *out, const float *bias, int KS, int64_t slice, int C, int64_t HW,
                                     int act, float alpha, float beta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slice; i += (int64_t)gridDim.x * blockDim.x) {
        float v = ws[i];
        for (int k = 1; k < KS; ++k) v += ws[i + k * slice];
        if (bias) v += bias[(int)((i / HW) % C)];
        out[i] = dvfp::apply_act(v, act, alpha, beta);
    }
}

// Split-K finish of a dgrad whose segment was produced by a ReLU layer: out = (mask > 0) ? sum_k ws[k] : 0 and
// dbias[c] += sum(out).  One block per chunk of one (n, c) plane: fixed-order sum of the partial tiles, one atomic per block.
__global__ __launch_bounds__(256) void splitk_reduce_mask_kernel(const float *ws, float *out, const float *mask, float *dbias,
                                                                 int KS, int64_t slice, int C, int HW, int chunks) {
    __shared__ float red[4];
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const int per = (HW + chunks - 1) / chunks;
    const int lo = chunk * per, hi = min(HW, lo + per);
    const int64_t base = (int64_t)plane * HW;
    float s = 0.f;
    for (int e = lo + threadIdx.x; e < hi; e += 256) {
        const int64_t i = base + e;
        float v = ws[i];
        for (int k = 1; k < KS; ++k) v += ws[i + k * slice];
        v = mask[i] > 0.f ? v : 0.f;
        out[i] = v;
        s += v;
    }
    if (!dbias) return;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dbias + plane % C, red[0] + red[1] + red[2] + red[3]);
}

// KS >= 8 (the 2x7 ... 8x26 layers: 8-64 partial tiles of a few thousand elements each): the k range is split over the four
// waves of a block -- each wave sums a contiguous quarter of the partial tiles in order, the quarters are added in order (a
// fixed tree: deterministic) -- and a block takes 64 consecutive elements, so the reduce of a 28 672-element output with
// KS = 64 is 448 blocks x 16 independent loads per thread instead of 112 blocks x 64 (round 3: 20 us -> the launch floor).
// mask != NULL: the dgrad finish of splitk_reduce_mask_kernel (out = mask > 0 ? sum : 0, dbias[c] += plane sums).
__global__ __launch_bounds__(256) void splitk_reduce_kpar_kernel(const float *ws, float *out, const float *bias, const float *mask,
                                                                 float *dbias, int KS, int64_t slice, int C, int HW, int act,
                                                                 float alpha, float beta) {
    __shared__ float part[3][64];
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int k0 = (KS * kq) >> 2, k1 = (KS * (kq + 1)) >> 2;
    const int64_t i0 = (int64_t)blockIdx.x * 64, i = i0 + lane;
    float v = 0.f;
    if (i < slice) {
        const float *p = ws + i + (int64_t)k0 * slice;
#pragma unroll 8
        for (int k = k0; k < k1; ++k, p += slice) v += *p;
    }
    if (kq) part[kq - 1][lane] = v;
    __syncthreads();
    if (kq) return;
    const bool ok = i < slice;
    v = ((v + part[0][lane]) + part[1][lane]) + part[2][lane];
    const int64_t plane = ok ? i / HW : -1;
    if (!mask) {
        if (ok) {
            if (bias) v += bias[(int)(plane % C)];
            out[i] = dvfp::apply_act(v, act, alpha, beta);
        }
        return;
    }
    v = (ok && mask[i] > 0.f) ? v : 0.f;
    if (ok) out[i] = v;
    if (!dbias) return;
    // the 64 elements of this wave span a few (n, c) planes: one atomic per plane
    const int64_t ilast = (i0 + 63 < slice ? i0 + 63 : slice - 1);
    for (int64_t pl = i0 / HW; pl <= ilast / HW; ++pl) {
        const float s = wave_sum(plane == pl ? v : 0.f);
        if (lane == 0) atomicAdd(dbias + (int)(pl % C), s);
    }
}

#ifdef DVF_TUNING
constexpr size_t STAMP_MAX_BLOCKS = 16384;
unsigned long long *g_stamp_buf = nullptr;
int g_stamp_blocks = 0;
#endif
// ---------------------------------------------------------------------------------------- host planning
struct PipePlan {
    int MT, NT, WM, CKH, TBU, threads;
    size_t lds;
    dim3 grid;
    int64_t packed_floats, ws_floats;
};

struct PipeOverride { int on, MT, NT, WM, CK, KS, BN, NST; };
inline PipeOverride pipe_override() {
    PipeOverride o{0, 0, 0, 0, 0, 0, 0, 0};
    if (const char *e = dvf_tune("DVF_PIPE_PLAN"))         // "MT,NT,WM,CK,KS,BN,NST" (0 = automatic) -- tuning tool only
        if (sscanf(e, "%d,%d,%d,%d,%d,%d,%d", &o.MT, &o.NT, &o.WM, &o.CK, &o.KS, &o.BN, &o.NST) >= 1) o.on = 1;
    return o;
}

// tile arrangement for `ntile` 32-pixel tiles per block over BN images
struct PipeTiles { int lsw, lsh, TGX, TGY, TGN, BN; double cov; };
inline PipeTiles pipe_tiles(int OHc, int OWc, int N, int ntile, int BNwant, int IS, int TBmax) {
    PipeTiles best{5, 0, 1, ntile, 1, 1, 1e30};
    for (int lsn = 0; lsn <= 5; ++lsn) {
        const int SN = 1 << lsn;
        for (int lsw = 5 - lsn; lsw >= 0; --lsw) {
            const int lsh = 5 - lsn - lsw, SW = 1 << lsw, SH = 1 << lsh;
            for (int TGN = 1; TGN <= ntile; TGN *= 2) {
                const int BN = SN * TGN;
                if (BN != BNwant) continue;
                for (int TGX = 1; TGX * TGN <= ntile; TGX *= 2) {
                    const int TGY = ntile / (TGX * TGN), BW = TGX * SW, BH = TGY * SH;
                    if ((BW - 1) * IS + TBmax > 160) continue;
                    const double cov = (double)cdiv(OWc, BW) * BW * cdiv(OHc, BH) * BH * cdiv(N, BN) * BN;
                    // narrow sub-tiles store / gather in short segments: prefer >= 16 pixels per row
                    // and >= 32 per block row (full 128-byte lines per store)
                    const double narrow = (SW >= 16 ? 0.0 : (SW == 8 ? 0.12 : 0.3)) + (BW >= 32 ? 0.0 : 0.08);
                    const double cost = cov * (1.0 + narrow + 0.02 * lsh + 0.02 * lsn) + 1e-3 * (BW + BH);
                    if (cost < best.cov) best = PipeTiles{lsw, lsh, TGX, TGY, TGN, BN, cost};
                }
            }
        }
    }
    return best;
}


// Plan one op (all its classes share the launch).  Fills `a`'s tiling/packing fields; the same plan drives packing.
int plan_pipe(PipeArgs &a, const ClassSpec *cls, int ncls, PipePlan &pl, bool head_fwd = false) {
    if (ncls < 1 || ncls > 4) return DVF_ERR_INVALID_ARG;
    a.ncls = ncls; a.OS = cls[0].OS; a.IS = cls[0].IS;
    int Tmax = 0, TAmax = 0, TBmax = 0, OHc = 0, OWc = 0;
    for (int i = 0; i < ncls; ++i) {
        const ClassSpec &c = cls[i];
        const int T = c.TA * c.TB;
        if (T < 1 || c.OHc <= 0 || c.OWc <= 0) return DVF_ERR_INVALID_ARG;
        Tmax = T > Tmax ? T : Tmax; TAmax = c.TA > TAmax ? c.TA : TAmax; TBmax = c.TB > TBmax ? c.TB : TBmax;
        OHc = c.OHc > OHc ? c.OHc : OHc; OWc = c.OWc > OWc ? c.OWc : OWc;
    }
    const PipeOverride ov = pipe_override();
    // Layers with at most 16 output channels stay on the register-staged conv_gather_kernel (half of a 32-row MFMA tile would
    // be padding); 17..32 channels run here since round 3 (with the rewritten producers the pipelined kernel wins for the 3x3
    // layers too: 3x3 65->32 @128x416 forward 0.143 -> 0.121 ms in the step).
    static const int smallm = dvf_tune("DVF_PIPE_SMALLM") ? atoi(dvf_tune("DVF_PIPE_SMALLM")) : 0;      // tuning knob
    // ... except the 1-4 channel heads over >= 128 input channels (K >= 1152 on a 32x104 map: the direct kernel is a chain of
    // 128 per-channel LDS round trips on 64 blocks -- 36-46 us -- against ~1 GFLOP of mostly-padding MFMA work here)
    int ctot = 0;
    for (int s = 0; s < a.nseg; ++s) ctot += a.segC[s];
    static const int headc = dvf_tune("DVF_PIPE_HEADC") ? atoi(dvf_tune("DVF_PIPE_HEADC")) : 128;       // tuning knob
    const bool wide_head = head_fwd && a.M <= 4 && ncls == 1 && ctot >= headc && Tmax == 9;
    // ... and the strided classes (dgrad of the pose network's 5x5 stride-2 16->32 layer: 85 us on the gather kernel, 59 here)
    const bool small_ok = a.M > 16 || wide_head || (!(smallm & 8) && a.M <= 16 && ncls > 1) ||
                          ((smallm & 2) && a.M <= 16) || ((smallm & 4) && a.M > 16);
    if (a.M <= 32 && !ov.on && !small_ok) return DVF_ERR_UNSUPPORTED;
    if (ov.on && ov.KS < 0) return DVF_ERR_UNSUPPORTED;      // (tuning: force the gather kernel)
    int maxc = 0;
    for (int s = 0; s < a.nseg; ++s) maxc = a.segC[s] > maxc ? a.segC[s] : maxc;
    const int64_t px = (int64_t)OHc * OWc;                   // pixels per image per class
    // --- taps per unit = the widest kernel row of the classes (narrower rows are zero-padded in the packed weights)
    int TBU = TBmax;
    if (TBU == 6) TBU = 7;
    if (TBU > 7) return DVF_ERR_UNSUPPORTED;
    (void)Tmax;
    // --- block shape: channels per block (32*MT*WM) and pixel tiles per block (WN*NT)
    int MT = a.M > 32 ? 2 : 1;
    int BN = 1;
    if (px <= 64) { while (BN < a.N && BN * px < 128 && BN < 32) BN *= 2; }
    int WM = 1;
    if (px * BN <= 64 && a.M >= 128) WM = 2;                 // two 32-pixel tiles per block at most: split waves over M
    int NT = 2;
    if (ov.on) {
        if (ov.MT) MT = ov.MT;
        if (ov.WM) WM = ov.WM;
        if (ov.BN) BN = ov.BN;
    }
    if (WM == 2 || MT == 2) NT = 1;                        // (64 channels x 2 tiles per wave does not fit the register file)
    const int WN = 4 / WM;
    auto nblocks = [&](const PipeTiles &tp) {
        return (int64_t)cdiv(OWc, tp.TGX << tp.lsw) * cdiv(OHc, tp.TGY << tp.lsh) * cdiv(a.M, 32 * MT * WM) *
               cdiv(a.N, tp.BN) * ncls;
    };
    PipeTiles tp = pipe_tiles(OHc, OWc, a.N, WN * NT, BN, a.IS, TBmax);
    if (tp.cov >= 1e30) return DVF_ERR_UNSUPPORTED;
    int64_t nblk = nblocks(tp);
    if (NT == 2 && nblk < 512 && !(ov.on && ov.NT)) {
        PipeTiles tp1 = pipe_tiles(OHc, OWc, a.N, WN * 1, BN, a.IS, TBmax);
        if (tp1.cov < 1e30) { NT = 1; tp = tp1; nblk = nblocks(tp); }
    }
    if (ov.on && ov.NT && WM == 1 && MT == 1) {
        NT = ov.NT;
        tp = pipe_tiles(OHc, OWc, a.N, WN * NT, BN, a.IS, TBmax);
        if (tp.cov >= 1e30) return DVF_ERR_UNSUPPORTED;
        nblk = nblocks(tp);
    }
    a.lsw = tp.lsw; a.lsh = tp.lsh; a.TGX = tp.TGX; a.TGY = tp.TGY; a.BN = tp.BN;
    a.BW = tp.TGX << tp.lsw; a.BH = tp.TGY << tp.lsh;
    a.tilesX = cdiv(OWc, a.BW); a.tilesY = cdiv(OHc, a.BH); a.NG = cdiv(a.N, a.BN);
    const int SW = 1 << a.lsw, SH = 1 << a.lsh, MB = 32 * MT * WM;
    // patch DMA in 16-byte lanes (pipe_geo): image rows and tile origins must be 16-byte aligned (the tensor base pointers
    // are checked at launch, pipe_run).  The wider rows cost LDS: where they would push the chunk depth below what 4-byte
    // lanes allow, the deeper chunks win (profiles/r03_pipe_stamps_after.txt: 3x3 128->128 @32x104, CK 16 vs 8).
    auto slab = [&](int CK, int TA) { return (TA * TBU * CK * MB + 255) & ~255; };
    static const int ks_thr = dvf_tune("DVF_PIPE_KSTHR") ? atoi(dvf_tune("DVF_PIPE_KSTHR")) : 160;       // tuning knobs
    static const int ks_tgt = dvf_tune("DVF_PIPE_KSTGT") ? atoi(dvf_tune("DVF_PIPE_KSTGT")) : 256;
    const int KS0 = nblk < ks_thr ? (int)(ks_tgt / nblk) : 1;     // split-K factor before clamping to the chunk count
    static const int big_lds_kb = dvf_tune("DVF_PIPE_BIGLDS_KB") ? atoi(dvf_tune("DVF_PIPE_BIGLDS_KB")) : 150;   // tuning knob
    // LDS budget: a grid of at most one block per CU may take (nearly) the whole 160 KiB; otherwise leave room for two
    const size_t PIPE_LDS_BUDGET = (nblk * KS0 <= 256 ? big_lds_kb : 76) * 1024;
    struct Depth { int ok, CK, NST, PSRmax; size_t lds; };
    auto depth_for = [&](int x4) {
        Depth d{0, 0, 3, 0, 0};
        for (int i = 0; i < ncls; ++i) {
            const PipeGeo g = pipe_geo(a.BH, a.BW, a.BN, a.IS, cls[i].TA, TBU, SW, SH, x4, cls[i].bx);
            if (g.NPI > PIPE_MAXNPI) return d;
            d.PSRmax = g.PSR > d.PSRmax ? g.PSR : d.PSRmax;
        }
        auto lds_bytes = [&](int CK) { return ((size_t)d.NST * (slab(CK, TAmax) + (size_t)CK * d.PSRmax) + MB) * 4; };
        // chunk depth: largest CK in {16,8,4} whose stages fit the LDS budget
        int CK = TBU >= 5 ? 8 : 16;                          // (the 5- and 7-tap kernels are only built for CK <= 8)
        while (CK > 4 && lds_bytes(CK) > PIPE_LDS_BUDGET) CK >>= 1;
        while (CK > 4 && CK / 2 >= maxc) CK >>= 1;
        if (lds_bytes(CK) > PIPE_LDS_BUDGET) {               // large kernels (5x5, 7x7): two stages, then one block per CU
            d.NST = 2;
            if (lds_bytes(CK) > PIPE_LDS_BUDGET && PIPE_LDS_BUDGET < 150 * 1024) {
                d.NST = 3;
                if (lds_bytes(CK) > 150 * 1024) d.NST = 2;
                if (lds_bytes(CK) > 150 * 1024) return d;
            }
        }
        if (ov.on && ov.CK) { CK = ov.CK; d.NST = 3; if (lds_bytes(CK) > 150 * 1024) d.NST = 2; }
        if (ov.on && ov.NST) d.NST = ov.NST;
        if ((CK != 4 && CK != 8 && CK != 16) || lds_bytes(CK) > 150 * 1024) return d;
        d.ok = 1; d.CK = CK; d.lds = lds_bytes(CK);
        return d;
    };
    int want_x4 = (a.IW % 4 == 0 && (a.BW * a.IS) % 4 == 0) ? 1 : 0;
    if (const char *e = dvf_tune("DVF_PIPE_X4")) want_x4 = want_x4 && atoi(e) != 0;    // tuning knob: 0 = 4-byte lanes everywhere
    const Depth d1 = depth_for(0), d4 = want_x4 ? depth_for(1) : Depth{0, 0, 3, 0, 0};
    static const bool x4_force = dvf_tune("DVF_PIPE_X4") && atoi(dvf_tune("DVF_PIPE_X4")) == 2;           // tuning: 2 = whenever legal
    const bool use4 = d4.ok && (!d1.ok || x4_force || (d4.CK >= d1.CK && d4.NST >= d1.NST));
    const Depth dd = use4 ? d4 : d1;
    if (!dd.ok) return DVF_ERR_UNSUPPORTED;
    a.x4 = use4 ? 1 : 0;
    a.PSRmax = dd.PSRmax;
    const int CK = dd.CK, NST = dd.NST;
    a.NST = NST;
    a.SLmax = slab(CK, TAmax);
    a.NCH = 0;
    for (int s = 0; s < a.nseg; ++s) a.NCH += cdiv(a.segC[s], CK);
    const int mblocks = cdiv(a.M, MB);
    unsigned off = 0;
    for (int i = 0; i < ncls; ++i) {
        const ClassSpec &c = cls[i];
        const int SL = slab(CK, c.TA);
        a.cls[i] = PClass{c.py, c.px, c.by, c.bx, c.TA, c.TB, c.OHc, c.OWc, SL, c.TA, off};
        const int64_t nf = (int64_t)mblocks * a.NCH * SL;
        if ((int64_t)off + nf >= ((int64_t)1 << 29)) return DVF_ERR_UNSUPPORTED;      // 32-bit byte offsets
        off += (unsigned)nf;
    }
    pl.packed_floats = off;
    // --- split-K: only when the grid leaves most CUs idle (one block per CU already pipelines loads under MFMAs)
    int KS = KS0;
    if (ov.on && ov.KS) KS = ov.KS;
    if (KS > a.NCH) KS = a.NCH;
    if (KS < 1) KS = 1;
    if ((int64_t)a.NG * KS * ncls > 65535) return DVF_ERR_UNSUPPORTED;
    a.KS = KS;
    pl.ws_floats = KS > 1 ? (int64_t)KS * a.N * a.M * a.OH * a.OW : 0;
    for (int s2 = 0; s2 < a.nseg; ++s2)
        if ((int64_t)a.N * a.segC[s2] * a.IH * a.IW * 4 >= ((int64_t)1 << 31) - 16) return DVF_ERR_UNSUPPORTED;
    pl.MT = MT; pl.NT = NT; pl.WM = WM; pl.CKH = CK / 2; pl.TBU = TBU; pl.lds = dd.lds;
    pl.grid = dim3(a.tilesX * a.tilesY, mblocks, a.NG * KS * ncls);
    // Blocked accumulation (conv_pipe.h): the accumulators are banked every `blk` chunks.  Product policy = round 2/3's: every
    // chunk, in the deep layers only (64-channel waves, 8/16-channel chunks of 2-4-tap rows, a K range of >= 512 terms on a
    // block that owns its CU) -- the layers whose sequential 2304-9216-term chains were up to 5.8x further from fp64 than the
    // CPU reference.  Banking EVERYWHERE (~96 terms apart: tuning knob DVF_PIPE_BLKT) brings every layer's rms error to
    // 1.4e-7, below torch's CPU kernels (1.7e-7; iconv3's unblocked 1161-term chain: 4.3e-7), but it also moves the results
    // AWAY from the reference's own rounding: torch's CPU convolution accumulates K sequentially, the unblocked kernels
    // reproduce its cfg-1 gradients to 2e-5 where the blocked ones differ by the fp32 noise floor (1e-4 in the deep layers),
    // which flips the sign of 162 of 12 M first Adam steps and moves the second iteration's loss by 0.5 % (tools/r3/
    // cfg1_flips.py, profiles/r03_blocked_accumulation.txt).  Parity with the reference wins; the knob stays in the tuning build.
    {
        const int per_chunk = CK * TAmax * TBmax;
        const int64_t kseq = (int64_t)cdiv(a.NCH, KS) * per_chunk;
        static const int blk_knob = dvf_tune("DVF_PIPE_BLK") ? atoi(dvf_tune("DVF_PIPE_BLK")) : -1;     // tuning: 0 off, n: every n chunks
        static const int blk_terms = dvf_tune("DVF_PIPE_BLKT") ? atoi(dvf_tune("DVF_PIPE_BLKT")) : 0;   // tuning: terms per bank, all layers
        const bool deep = MT == 2 && NT == 1 && WM == 1 && CK >= 8 && TBU >= 2 && TBU <= 4 && kseq >= 512 &&
                          ((int64_t)pl.grid.x * pl.grid.y * pl.grid.z <= 256 || pl.lds > 76 * 1024);
        int every = blk_terms > 0 ? (blk_terms + per_chunk / 2) / per_chunk : 1;
        if (every < 1) every = 1;
        a.blk = blk_knob == 0 ? 0 : blk_knob > 0 ? blk_knob : blk_terms > 0 ? (kseq >= 384 ? every : 0) : (deep ? 1 : 0);
    }
    // a block that has a CU to itself (by LDS size or by grid size) gets four producer waves, one per SIMD
    {
        const int64_t nblocks_total = (int64_t)pl.grid.x * pl.grid.y * pl.grid.z;
        static const bool no4p = dvf_tune("DVF_PIPE_NO4P") != nullptr;
        // (32-channel MFMA waves need <= 128 VGPRs: two 8-wave blocks still fit a CU)
        static const int mt1_4p = dvf_tune("DVF_PIPE_MT1_4P") ? atoi(dvf_tune("DVF_PIPE_MT1_4P")) : 1;      // tuning knob
        pl.threads = (!no4p && (pl.lds > 76 * 1024 || nblocks_total <= 256 || (MT == 1 && mt1_4p))) ? dvfp::PIPE_THREADS_4P : dvfp::PIPE_THREADS;
    }
    return DVF_OK;
}

int launch_pipe(const PipeArgs &a, const PipePlan &pl, hipStream_t st) {
    using dvfp::launch_pipe_family;
    if (pl.WM == 2) {
        if (pl.NT != 1) return DVF_ERR_UNSUPPORTED;
        return pl.MT == 2 ? launch_pipe_family<2, 1, 2>(a, pl.CKH, pl.TBU, pl.grid, pl.lds, st, pl.threads)
                          : launch_pipe_family<1, 1, 2>(a, pl.CKH, pl.TBU, pl.grid, pl.lds, st, pl.threads);
    }
    if (pl.WM != 1) return DVF_ERR_UNSUPPORTED;
    if (pl.MT == 2 && pl.NT == 2) return DVF_ERR_UNSUPPORTED;
    if (pl.MT == 2) return launch_pipe_family<2, 1, 1>(a, pl.CKH, pl.TBU, pl.grid, pl.lds, st, pl.threads);
    if (pl.NT == 2) return launch_pipe_family<1, 2, 1>(a, pl.CKH, pl.TBU, pl.grid, pl.lds, st, pl.threads);
    return launch_pipe_family<1, 1, 1>(a, pl.CKH, pl.TBU, pl.grid, pl.lds, st, pl.threads);
}

// One op instance = one (M range, reduction segments, classes) triple: Conv2d fwd, one segment of a dgrad, ...
struct PipeOp {
    PipeArgs a;
    ClassSpec cls[4];
    int ncls;
    bool covers;
    int w_mode, Mtot, Rtot, KK, m_base;
    bool head_fwd;       // forward of a Conv2d (make_fwd_op): the 1-4 channel heads over >= 128 channels may run here
};

// ws: optional split-K workspace (ws_floats floats).  With it the K-splits store plain partial tiles and one pass
// reduces them (+bias, activation); without it they accumulate with float atomics into a zeroed output.
int pipe_run(PipeOp &op, const float *packed, float *ws, int64_t ws_floats, hipStream_t st) {
    PipePlan pl;
    int rc = plan_pipe(op.a, op.cls, op.ncls, pl, op.head_fwd);
    if (rc) return rc;
    PipeArgs &a = op.a;
    a.wp = packed;
    const int64_t HW = (int64_t)a.OH * a.OW, total = (int64_t)a.N * a.M * HW;
    float *out = a.out;
    int mode = 0;
    if (!op.covers) mode = 1;
    else if (a.KS > 1) mode = (ws && ws_floats >= pl.ws_floats) ? 2 : 1;
    if (mode == 1 && hipMemsetAsync(out, 0, sizeof(float) * total, st) != hipSuccess) return DVF_ERR_LAUNCH;
    a.out_mode = mode;
    if (mode == 2) { a.out = ws; a.ws_slice = total; }
    if (const char *e = dvf_tune("DVF_DBG")) a.dbg = atoi(e);
    a.stamps = nullptr;
#ifdef DVF_TUNING
    if (dvf_tune("DVF_STAMPS")) {      // in-kernel cycle account of the LAST pipelined launch (tools/r3/stamps.py)
        const size_t nb = (size_t)pl.grid.x * pl.grid.y * pl.grid.z;
        if (!g_stamp_buf && hipMalloc(&g_stamp_buf, STAMP_MAX_BLOCKS * 64) != hipSuccess) return DVF_ERR_LAUNCH;
        if (nb <= STAMP_MAX_BLOCKS) {
            if (hipMemsetAsync(g_stamp_buf, 0, nb * 64, st) != hipSuccess) return DVF_ERR_LAUNCH;
            a.stamps = g_stamp_buf;
            g_stamp_blocks = (int)nb;
        }
    }
#endif
    if (a.x4)
        for (int s2 = 0; s2 < a.nseg; ++s2)
            if (reinterpret_cast<uintptr_t>(a.in[s2]) & 15) return DVF_ERR_UNSUPPORTED;    // (a view at an odd element offset)
    dvf_plan_note(DVF_K_PIPE, pl.MT, pl.NT, pl.WM, 2 * pl.CKH, pl.TBU, a.KS, a.BN, a.NST, pl.threads, (int)pl.lds, mode | (a.ncls << 4) | (a.x4 << 8) | ((a.blk ? 1 : 0) << 9));
    if (dvf_tune("DVF_PIPE_DEBUG"))
        fprintf(stderr, "[pipe] M %d chunks %d N %d out %dx%d cls %d | MT %d NT %d WM %d CK %d TBU %d KS %d BN %d tile %dx%d "
                "(sub %dx%d) grid %ux%ux%u lds %zu x%d mode %d\n", a.M, a.NCH, a.N, a.OH, a.OW, a.ncls, pl.MT, pl.NT, pl.WM,
                2 * pl.CKH, pl.TBU, a.KS, a.BN, a.BH, a.BW, 1 << a.lsh, 1 << a.lsw, pl.grid.x, pl.grid.y, pl.grid.z, pl.lds, a.NST, mode);
    rc = launch_pipe(a, pl, st);
    if (rc) return rc;
    const int64_t nb = (total + 255) / 256;
    if (mode == 2 && a.KS >= 8 && HW < ((int64_t)1 << 30)) {
        splitk_reduce_kpar_kernel<<<(unsigned)((total + 63) / 64), 256, 0, st>>>(ws, out, a.bias, a.mask, a.dbias, a.KS, total, a.M,
                                                                                (int)HW, a.act, a.alpha, a.beta);
        DVF_LAUNCH_CHECK();
    } else if (mode == 2 && a.mask) {
        int chunks = 1;
        const int64_t planes = (int64_t)a.N * a.M;
        while (planes * chunks < 1024 && HW / (chunks * 2) >= 256) chunks *= 2;
        splitk_reduce_mask_kernel<<<(unsigned)(planes * chunks), 256, 0, st>>>(ws, out, a.mask, a.dbias, a.KS, total, a.M, (int)HW, chunks);
        DVF_LAUNCH_CHECK();
    } else if (mode == 2) {
        splitk_reduce_kernel<<<(int)(nb > 4096 ? 4096 : nb), 256, 0, st>>>(ws, out, a.bias, a.KS, total, a.M, HW, a.act,
                                                                          a.alpha, a.beta);
        DVF_LAUNCH_CHECK();
    } else if (mode == 1 && (a.bias || a.act != DVF_ACT_NONE)) {
        bias_act_kernel<<<(int)(nb > 2048 ? 2048 : nb), 256, 0, st>>>(out, a.bias, a.M, HW, total, a.act, a.alpha, a.beta);
        DVF_LAUNCH_CHECK();
    }
    if (mode == 1 && a.mask)    // atomic accumulation into a zeroed output: the mask pass runs over the finished sums
        return dvf_act_bwd2(out, a.mask, out, a.dbias, a.N, a.M, (int)HW, DVF_ACT_RELU, 1.f, 0.f, 1, st);
    return DVF_OK;
}

int pipe_pack(PipeOp &op, const float *w, float *packed, int64_t *nfloats, int64_t *wsfloats, hipStream_t st,
              PackArgs *job_out = nullptr, int *job_blocks = nullptr) {
    PipePlan pl;
    int rc = plan_pipe(op.a, op.cls, op.ncls, pl, op.head_fwd);
    if (rc) return rc;
    if ((size_t)32 * ((2 * pl.CKH * op.KK) | 1) * sizeof(float) > 64 * 1024) return DVF_ERR_UNSUPPORTED;   // pack kernel's LDS tile
    if (nfloats) *nfloats = pl.packed_floats;
    if (wsfloats) *wsfloats = op.covers ? pl.ws_floats : 0;
    if (!w || !packed) return DVF_OK;                       // size query only
    const PipeArgs &a = op.a;
    PackArgs p{};
    p.w = w; p.wp = packed; p.w_mode = op.w_mode; p.Mtot = op.Mtot; p.Rtot = op.Rtot; p.KK = op.KK;
    p.m_base = op.m_base; p.M = a.M; p.nseg = a.nseg;
    for (int s = 0; s < a.nseg; ++s) p.segC[s] = a.segC[s];
    p.ncls = op.ncls; p.NCH = a.NCH; p.CK = 2 * pl.CKH; p.MB = 32 * pl.MT * pl.WM; p.VW = pl.CKH >= 4 ? 4 : pl.CKH;
    p.TBU = pl.TBU;
    for (int c = 0; c < op.ncls; ++c) {
        p.SL[c] = a.cls[c].SL;
        p.TB[c] = op.cls[c].TB;
        p.TA[c] = op.cls[c].TA;
        p.wp_off[c] = a.cls[c].wp_off;
        for (int t = 0; t < op.cls[c].TA * op.cls[c].TB; ++t) p.tapmap[c][t] = (signed char)op.cls[c].tapmap[t];
    }
    const size_t lds = (size_t)32 * ((p.CK * p.KK) | 1) * sizeof(float);
    if (lds > 64 * 1024) return DVF_ERR_UNSUPPORTED;
    const int mtiles = cdiv(a.M, p.MB) * (p.MB >> 5);
    if (job_out) {                                          // export for dvf_conv2d_pack_batch instead of launching
        *job_out = p;
        *job_blocks = a.NCH * mtiles;
        return DVF_OK;
    }
    conv_pack_kernel<<<dim3(a.NCH, mtiles), 256, lds, st>>>(p);
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}
