// Direct (non-MFMA) kernels for the 1-4 channel 3x3 prediction heads (predict_disp*, DispNetS.py:66-69; the
// explainability-mask heads, PoseExpNet_sfm.py:43-46).  A 32-row MFMA tile would be 88-97 % padding for these
// layers; they are memory-bound (read the C_in-channel activation once), so plain FMAs over an LDS-staged tile are
// the right tool: forward, dgrad and wgrad.  3x3, stride 1, padding 1, single input segment, C_out <= 4.
#include "conv_head.h"

namespace {

constexpr int HT_W = 64, HT_H = 4;          // pixel tile per block (256 threads, one pixel each)
constexpr int HP_W = HT_W + 2, HP_H = HT_H + 2;
constexpr int HCK = 8;                      // input channels staged per pass (forward)
constexpr int HWK = 7;                      // ... in wgrad: 7 channels x 9 taps = 63 columns, one per lane

__device__ __forceinline__ float head_act(float v, int act, float alpha, float beta) {
    if (act == DVF_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DVF_ACT_SIGMOID_AFFINE) return alpha * (1.f / (1.f + expf(-v))) + beta;
    return v;
}

// stage `nch` channel planes (tile + halo 1, zero outside the image) into lds[ch][HP_H][HP_W]
__device__ __forceinline__ void stage_tile(float *lds, const float *__restrict__ src, int64_t plane, int nch, int H, int W,
                                           int y0, int x0) {
    for (int e = threadIdx.x; e < nch * HP_H * HP_W; e += 256) {
        const int ch = e / (HP_H * HP_W), r = e - ch * (HP_H * HP_W), py = r / HP_W, px = r - py * HP_W;
        const int y = y0 + py - 1, x = x0 + px - 1;
        lds[e] = (y >= 0 && y < H && x >= 0 && x < W) ? src[ch * plane + (int64_t)y * W + x] : 0.f;
    }
}

// out[n][mo][y][x] = act(b[mo] + sum_c sum_taps w[mo][c][tap] * in[n][c][y+ta-1][x+tb-1])
// Weight addressing is general: w[m * ws_m + c * ws_c + tap'] with tap' = flip ? 8 - tap : tap, so the same kernel also
// computes the dgrad of a narrow input segment (m = segment channel, c = output channel of the convolution, taps flipped).
// Tile 64 x 8 pixels, two rows per thread; the per-thread staging offsets are channel-invariant.
constexpr int FT_H = 8;                    // tile rows of the full-resolution variant (head_fwd_kernel<.., 8, 8>)
constexpr int HEAD_MAX_SEGS = 3;             // (the thin layers these kernels serve concatenate at most three tensors)
struct HeadSegs {                            // the input: a virtual concat of up to HEAD_MAX_SEGS tensors
    const float *p[HEAD_MAX_SEGS];
    int c[HEAD_MAX_SEGS];
    int n;
};
// HCKT input channels per staging round, FTH tile rows (8: two rows per thread; 4: one).  Full-resolution layers run
// <8, 8>; the low-resolution heads (<= 64x208: a few dozen blocks, whose run time is the chain of their staging round
// trips -- 16 rounds for the 128-channel head) run <32, 4>: four times fewer, four times larger rounds on twice the blocks.
template <int MO, int HCKT, int FTH>
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadSegs in, const float *__restrict__ w,
                                                       const float *__restrict__ bias, float *__restrict__ out, int C, int H,
                                                       int W, int tilesX, int act, float alpha, float beta, int ws_m, int ws_c,
                                                       int flip, const float *__restrict__ mask) {
    constexpr int ROWS = FTH / 4, FPH = FTH + 2, FPN = FPH * HP_W, FLD = (FPN + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float wsh[];           // [MO][C][9] weights of the block (tap order already flipped if asked) | tile
    float *tile = wsh + ((MO * C * 9 + 3) & ~3);                          // [HCKT][FPN]
    // LDS layout: [m][c][9] for the 1-4 channel heads; [c][m][9] for the 16-channel variant, whose inner loop reads the 144
    // weights of a channel as 36 uniform-address ds_read_b128 (one broadcast read per four FMAs pairs instead of one per pair)
    for (int e = threadIdx.x; e < MO * C * 9; e += 256) {
        const int m = e / (C * 9), r = e - m * (C * 9), c = r / 9, k = r - c * 9;
        const float wv = w[(int64_t)m * ws_m + (int64_t)c * ws_c + (flip ? 8 - k : k)];
        if (MO == 16) wsh[(c * MO + m) * 9 + k] = wv;
        else wsh[e] = wv;
    }
    const int n = blockIdx.y, tY = blockIdx.x / tilesX, tX = blockIdx.x - tY * tilesX;
    const int y0 = tY * FTH, x0 = tX * HT_W;
    const int lx = threadIdx.x & (HT_W - 1), ly = (threadIdx.x >> 6) * ROWS;   // rows ly .. ly+ROWS-1
    const int64_t plane = (int64_t)H * W;
    int soff[FLD];                           // source offset inside a plane (or -1: zero / not mine)
#pragma unroll
    for (int i = 0; i < FLD; ++i) {
        const int e = threadIdx.x + 256 * i, py = e / HP_W, px = e - py * HP_W;
        const int y = y0 + py - 1, x = x0 + px - 1;
        soff[i] = (e < FPN && y >= 0 && y < H && x >= 0 && x < W) ? y * W + x : -1;
    }
    float acc[ROWS][MO];
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r][m] = bias ? bias[m] : 0.f;
    // the loads of chunk c0 + HCKT are issued before the FMAs of chunk c0 (registers), so their latency overlaps them
    float stg[HCKT][FLD];
    static_assert(HEAD_MAX_SEGS == 3, "segment select below");
    auto seg_ptr = [&](int sg) { return sg == 0 ? in.p[0] : sg == 1 ? in.p[1] : in.p[2]; };
    auto seg_ch = [&](int sg) { return sg == 0 ? in.c[0] : sg == 1 ? in.c[1] : in.c[2]; };
    auto issue = [&](int sg, int cs) {       // chunk = up to HCKT channels of ONE segment, starting at its channel cs
        const int sc = seg_ch(sg), nch = min(HCKT, sc - cs);
        const float *src = seg_ptr(sg) + ((int64_t)n * sc + cs) * plane;
#pragma unroll
        for (int ch = 0; ch < HCKT; ++ch)
#pragma unroll
            for (int i = 0; i < FLD; ++i) stg[ch][i] = (ch < nch && soff[i] >= 0) ? src[ch * plane + soff[i]] : 0.f;
    };
    issue(0, 0);
    int sg = 0, cs = 0, cg = 0;              // current chunk: segment, channel inside it, channel of the concat
    while (sg < in.n) {
        const int nch = min(HCKT, seg_ch(sg) - cs);
        int sg2 = sg, cs2 = cs + HCKT;
        if (cs2 >= seg_ch(sg)) { ++sg2; cs2 = 0; }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < HCKT; ++ch)
#pragma unroll
            for (int i = 0; i < FLD; ++i) {
                const int e = threadIdx.x + 256 * i;
                if (e < FPN) tile[ch * FPN + e] = stg[ch][i];
            }
        __syncthreads();
        if (sg2 < in.n) issue(sg2, cs2);
        // (1-4 channel heads) four input channels in flight with their own accumulators: a thread's work is one long
        // chain of LDS reads and dependent FMAs (288 per 32-channel round), and these layers run a block or less per CU
        if constexpr (MO != 16) {
            float part[3][ROWS][MO];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int m = 0; m < MO; ++m) part[q][r][m] = 0.f;
            int ch = 0;
            for (; ch + 4 <= nch; ch += 4) {
                float v4[4][ROWS + 2][3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float *t = tile + (ch + q) * FPN + ly * HP_W + lx;
#pragma unroll
                    for (int a = 0; a < ROWS + 2; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) v4[q][a][b] = t[a * HP_W + b];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int m = 0; m < MO; ++m) {
                        const float *wm = wsh + (m * C + cg + ch + q) * 9;
#pragma unroll
                        for (int k = 0; k < 9; ++k) {
                            const float wk = wm[k];
#pragma unroll
                            for (int r = 0; r < ROWS; ++r) {
                                if (q == 0) acc[r][m] = fmaf(wk, v4[q][k / 3 + r][k % 3], acc[r][m]);
                                else part[q - 1][r][m] = fmaf(wk, v4[q][k / 3 + r][k % 3], part[q - 1][r][m]);
                            }
                        }
                    }
            }
            for (; ch < nch; ++ch) {
                const float *t = tile + ch * FPN + ly * HP_W + lx;
#pragma unroll
                for (int m = 0; m < MO; ++m) {
                    const float *wm = wsh + (m * C + cg + ch) * 9;
#pragma unroll
                    for (int k = 0; k < 9; ++k)
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) acc[r][m] = fmaf(wm[k], t[(k / 3 + r) * HP_W + k % 3], acc[r][m]);
                }
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int m = 0; m < MO; ++m) acc[r][m] += (part[0][r][m] + part[1][r][m]) + part[2][r][m];
        }
        for (int ch = 0; ch < (MO == 16 ? nch : 0); ++ch) {
            const float *t = tile + ch * FPN + ly * HP_W + lx;
            float v[ROWS + 2][3];
#pragma unroll
            for (int a = 0; a < ROWS + 2; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) v[a][b] = t[a * HP_W + b];
            if constexpr (MO == 16) {
                typedef float f4h __attribute__((ext_vector_type(4)));
                const f4h *wq = reinterpret_cast<const f4h *>(wsh + (cg + ch) * (MO * 9));     // 16-byte aligned: 144 floats per channel
                f4h wr[MO * 9 / 4];
#pragma unroll
                for (int q = 0; q < MO * 9 / 4; ++q) wr[q] = wq[q];
#pragma unroll
                for (int m = 0; m < MO; ++m)
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const int f = m * 9 + k;
                        const float wk = wr[f >> 2][f & 3];
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) acc[r][m] = fmaf(wk, v[k / 3 + r][k % 3], acc[r][m]);
                    }
            } else {
#pragma unroll
                for (int m = 0; m < MO; ++m) {
                    const float *wm = wsh + (m * C + cg + ch) * 9;             // LDS broadcast reads
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const float wk = wm[k];
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) acc[r][m] = fmaf(wk, v[k / 3 + r][k % 3], acc[r][m]);
                    }
                }
            }
        }
        cg += nch; sg = sg2; cs = cs2;
    }
    const int x = x0 + lx;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int y = y0 + ly + r;
        if (y < H && x < W) {
            if (mask) {        // (segment dgrad) ReLU backward of the layer that produced this segment: its output is the mask
#pragma unroll
                for (int m = 0; m < MO; ++m) {
                    const int64_t idx = ((int64_t)n * MO + m) * plane + (int64_t)y * W + x;
                    out[idx] = mask[idx] > 0.f ? acc[r][m] : 0.f;
                }
            } else {
#pragma unroll
                for (int m = 0; m < MO; ++m)
                    out[((int64_t)n * MO + m) * plane + (int64_t)y * W + x] = head_act(acc[r][m], act, alpha, beta);
            }
        }
    }
}

// din[n][c][y][x] = sum_mo sum_taps w[mo][c][ta][tb] * dpre[n][mo][y-ta+1][x-tb+1]
template <int MO>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float *__restrict__ dpre, const float *__restrict__ w,
                                                         float *__restrict__ din, int C, int H, int W, int tilesX,
                                                         const float *__restrict__ mask) {
    __shared__ float tile[MO * HP_H * HP_W];
    extern __shared__ __attribute__((aligned(16))) float wsh[];           // [C][MO*9 (+pad to 4)]: the layer's weights, once per block
    // (the per-channel loop used to fetch its 9*MO weights with scalar loads from global memory: a dependent ~0.4 us round
    // trip per channel, 50 us for the 128-channel head at 32x104; from LDS they are broadcast reads)
    constexpr int WPC = (MO * 9 + 3) & ~3;
    for (int e = threadIdx.x; e < C * MO * 9; e += 256) {
        const int c = e / (MO * 9), r = e - c * (MO * 9), m = r / 9, k = r - m * 9;
        wsh[c * WPC + r] = w[((int64_t)m * C + c) * 9 + k];
    }
    const int n = blockIdx.y, tY = blockIdx.x / tilesX, tX = blockIdx.x - tY * tilesX;
    const int y0 = tY * HT_H, x0 = tX * HT_W;
    const int lx = threadIdx.x & (HT_W - 1), ly = threadIdx.x >> 6;
    const int64_t plane = (int64_t)H * W;
    stage_tile(tile, dpre + (int64_t)n * MO * plane, plane, MO, H, W, y0, x0);
    __syncthreads();
    float g[MO][9];                          // g[mo][ta*3+tb] = dpre[mo][y-ta+1][x-tb+1]
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[m][a * 3 + b] = tile[m * (HP_H * HP_W) + (ly + 2 - a) * HP_W + (lx + 2 - b)];
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) return;
    const int64_t base = (int64_t)n * C * plane + (int64_t)y * W + x;
    float *dst = din + base;
    const float *mk = mask ? mask + base : nullptr;     // ReLU backward of the layer that produced the input: its output is the mask
    // four channels per trip: their mask loads are issued together and the stores leave back to back
    for (int c0 = 0; c0 < C; c0 += 4) {
        float sv[4], mv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = min(c0 + q, C - 1);
            mv[q] = mk ? mk[c * plane] : 1.f;
            const float *wc = wsh + c * WPC;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MO; ++m)
#pragma unroll
                for (int k = 0; k < 9; ++k) s = fmaf(wc[m * 9 + k], g[m][k], s);
            sv[q] = s;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c0 + q < C) dst[(c0 + q) * plane] = (mv[q] > 0.f) ? sv[q] : 0.f;
    }
}

// dW[mo][c][ta][tb] (+)= sum_{n,y,x} dpre[n][mo][y][x] * in[n][c][y+ta-1][x+tb-1]
// grid (channel chunks of HWK, pixel-tile groups); thread t < nch*9 owns column (c, tap) and walks the tile's pixels
// part != NULL: the block's sums go to part[blockIdx.y][m][c][tap] (head_wgrad_finish_kernel adds the groups in order)
template <int MO>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float *__restrict__ in, const float *__restrict__ dpre,
                                                         float *dw, int N, int C, int H, int W, int tilesX, int tilesY,
                                                         float *part = nullptr) {
    __shared__ float tin[HWK * HP_H * HP_W];
    __shared__ float tdp[MO * HT_H * HT_W];
    __shared__ float red[4 * HWK * 9 * MO];
    const int c0 = blockIdx.x * HWK, nch = min(HWK, C - c0);
    const int64_t plane = (int64_t)H * W;
    // 4 waves split the tile's rows; lane < nch*9 owns one (c, tap) column
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col_c = lane / 9, col_t = lane - col_c * 9, ta = col_t / 3, tb = col_t - ta * 3;
    const bool active = lane < nch * 9;
    float acc[MO];
#pragma unroll
    for (int m = 0; m < MO; ++m) acc[m] = 0.f;
    const int ntiles = N * tilesX * tilesY;
    for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
        const int n = tile / (tilesX * tilesY), r = tile - n * (tilesX * tilesY), tY = r / tilesX, tX = r - tY * tilesX;
        const int y0 = tY * HT_H, x0 = tX * HT_W;
        __syncthreads();
        stage_tile(tin, in + ((int64_t)n * C + c0) * plane, plane, nch, H, W, y0, x0);
        for (int e = threadIdx.x; e < MO * HT_H * HT_W; e += 256) {
            const int m = e / (HT_H * HT_W), q = e - m * (HT_H * HT_W), py = q / HT_W, px = q - py * HT_W;
            const int y = y0 + py, x = x0 + px;
            tdp[e] = (y < H && x < W) ? dpre[((int64_t)n * MO + m) * plane + (int64_t)y * W + x] : 0.f;
        }
        __syncthreads();
        if (active) {
            const float *src = tin + col_c * (HP_H * HP_W) + (wave + ta) * HP_W + tb;      // row `wave` of the tile
            const float *dp = tdp + wave * HT_W;
#pragma unroll 8
            for (int px = 0; px < HT_W; ++px) {
                const float v = src[px];
#pragma unroll
                for (int m = 0; m < MO; ++m) acc[m] = fmaf(dp[m * (HT_H * HT_W) + px], v, acc[m]);     // broadcast reads
            }
        }
    }
    // combine the 4 waves, one atomic per output
    if (active)
#pragma unroll
        for (int m = 0; m < MO; ++m) red[(wave * MO + m) * (HWK * 9) + lane] = acc[m];
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int m = 0; m < MO; ++m) {
            const float s = (red[(0 * MO + m) * (HWK * 9) + lane] + red[(1 * MO + m) * (HWK * 9) + lane]) +
                            (red[(2 * MO + m) * (HWK * 9) + lane] + red[(3 * MO + m) * (HWK * 9) + lane]);
            const int64_t o = ((int64_t)m * C + c0 + col_c) * 9 + col_t;
            if (part) part[(int64_t)blockIdx.y * MO * C * 9 + o] = s;
            else atomicAdd(&dw[o], s);
        }
    }
}

// dw[e] += sum over groups g (in order) of part[g][e]
__global__ __launch_bounds__(256) void head_wgrad_finish_kernel(const float *__restrict__ part, float *dw, int n, int groups) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += part[(int64_t)g * n + e];
    dw[e] += s;
}


// ------------------------------------------------------------------------------------ thin stride-2 transposed convolutions
// ConvTranspose2d(k = 3, s = 2, p = 1, output_padding = 1) and (k = 4, s = 2, p = 1) with 16 output channels
// (DispNetS upconv1, DispNetS.py:62-64; PoseExpNet upconv1, PoseExpNet_sfm.py:38-41): the output is at or
// near the input resolution of the network, the reduction is 32-64 channels x 1-4 taps per output pixel.  As an MFMA GEMM
// that is four parity-class launches-in-one with 16 of 32 rows empty and a 32-256 deep K: the gather kernel ran them at
// 10-15 % of the matrix peak.  Direct form: one thread owns ONE INPUT pixel and its 2x2 output pixels x CO channels
// (4*CO accumulators); per input channel it reads the 2x2 (k = 3) or 3x3 (k = 4) input neighbourhood and does the
// 9*CO (16*CO) FMAs of that channel.  The weights of the layer (C*CO*k*k floats, <= 64 KB) are copied to LDS once per block
// and read back as uniform-address ds_read_b128 (broadcast: four weights per read, one read per four FMAs).  (Scalar loads
// + SGPR operands were tried first: 6-23 TF, the waves sit in s_waitcnt behind the weight stream.)  Layout: w[ci][co][a][b].
//   k = 3: out(2y+dy, 2x+dx) uses rows (a, iy) in {(1, y)} for dy = 0, {(0, y+1), (2, y)} for dy = 1      (Y = 2*iy - 1 + a)
//   k = 4: rows {(1, y), (3, y-1)} for dy = 0, {(0, y+1), (2, y)} for dy = 1;  columns likewise.
template <int CO, int K>
__global__ __launch_bounds__(256) void dconvt_s2_fwd_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ out, int C, int H,
                                                            int W, int OH, int OW, int act, float alpha, float beta) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];      // [C][CO*K*K]
    constexpr int WPC = CO * K * K;
    static_assert(WPC % 4 == 0, "weights of one input channel are read in 16-byte pieces");
    typedef float f4w __attribute__((ext_vector_type(4)));
    for (int e = threadIdx.x; e < C * WPC / 4; e += 256)
        reinterpret_cast<f4w *>(wlds)[e] = reinterpret_cast<const f4w *>(w)[e];
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
    const bool inside = x < W && y < H;
    // neighbourhood offsets: rows y-1, y, y+1 -> index 0, 1, 2 (k = 3 never uses index 0)
    const bool rok[3] = {y - 1 >= 0 && y - 1 < H, y < H, y + 1 < H}, cok[3] = {x - 1 >= 0 && x - 1 < W, x < W, x + 1 < W};
    float acc[2][2][CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float b = bias ? bias[co] : 0.f;
        acc[0][0][co] = acc[0][1][co] = acc[1][0][co] = acc[1][1][co] = b;
    }
    const int64_t plane = (int64_t)H * W;
    const float *ip = in + (int64_t)n * C * plane + (int64_t)y * W + x;
    for (int ci = 0; ci < C; ++ci) {
        float v[3][3];
#pragma unroll
        for (int r = (K == 3 ? 1 : 0); r < 3; ++r)
#pragma unroll
            for (int c = (K == 3 ? 1 : 0); c < 3; ++c) v[r][c] = (rok[r] && cok[c]) ? ip[(r - 1) * W + (c - 1)] : 0.f;
        const f4w *wc = reinterpret_cast<const f4w *>(wlds + ci * WPC);     // uniform address: LDS broadcast
        f4w wq[WPC / 4];
#pragma unroll
        for (int q = 0; q < WPC / 4; ++q) wq[q] = wc[q];
#pragma unroll
        for (int co = 0; co < CO; ++co) {
#pragma unroll
            for (int a = 0; a < K; ++a) {
                // output row parity and input row of tap a
                const int dy = (a + 1) & 1;                           // Y = 2*iy - 1 + a  ->  Y parity = (a + 1) & 1
                const int ry = 1 + ((dy + 1 - a) >> 1);             // iy - y + 1 = 1 + (dy + 1 - a) / 2  (exact: even numerator)
#pragma unroll
                for (int b = 0; b < K; ++b) {
                    const int dx = (b + 1) & 1;
                    const int rx = 1 + ((dx + 1 - b) >> 1);
                    const int f = co * K * K + a * K + b;
                    acc[dy][dx][co] = fmaf(wq[f >> 2][f & 3], v[ry][rx], acc[dy][dx][co]);
                }
            }
        }
        ip += plane;
    }
    if (!inside) return;
    const int64_t oplane = (int64_t)OH * OW;
    float *op = out + (int64_t)n * CO * oplane;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int Y = 2 * y + dy;
        if (Y >= OH) continue;
        const int X = 2 * x;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float r0 = head_act(acc[dy][0][co], act, alpha, beta), r1 = head_act(acc[dy][1][co], act, alpha, beta);
            float *q = op + co * oplane + (int64_t)Y * OW + X;
            if (X + 1 < OW && (OW & 1) == 0) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 t; t.x = r0; t.y = r1;
                *reinterpret_cast<f2 *>(q) = t;                   // (X even, OW even: 8-byte aligned)
            } else {
                if (X < OW) q[0] = r0;
                if (X + 1 < OW) q[1] = r1;
            }
        }
    }
}

inline int cdivh(int a, int b) { return (a + b - 1) / b; }

}  // namespace

bool dvf_head_applicable(const dvf_conv_desc *d, int nseg) {
    return nseg == 1 && !d->transposed && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->C_out >= 1 &&
           d->C_out <= 4 && d->H_out == d->H_in && d->W_out == d->W_in && d->C_in >= 4 && d->C_in * d->C_out <= 1024 &&
           dvf_tune("DVF_NO_HEAD") == nullptr;
}
// Per role (round 3, rocprofv3 durations per layer, profiles/r03_launch_table.txt): the direct kernels are a chain of
// per-channel LDS round trips per block, so the heads with many input channels on small maps (128 @32x104, 64 @64x208: a
// few dozen blocks) lose to conv_pipe_kernel even with 94-97 % of its MFMA rows padding:
//   forward 128->2 @32x104   46 us direct            dgrad 128->1 @32x104   48 us direct, 18 us conv_pipe
//                                                    dgrad  64->1 @64x208   28 us direct, 18 us conv_pipe
// The weight gradient stays direct at every size (13-58 us against 22-73 us).
bool dvf_head_fwd_applicable(const dvf_conv_desc *d, int nseg) {
    static const int maxc = dvf_tune("DVF_HEAD_FWD_MAXC") ? atoi(dvf_tune("DVF_HEAD_FWD_MAXC")) : 127;      // tuning knob
    return dvf_head_applicable(d, nseg) && d->C_in <= maxc;
}
bool dvf_head_dgrad_applicable(const dvf_conv_desc *d, int nseg) {
    static const int maxc = dvf_tune("DVF_HEAD_DGRAD_MAXC") ? atoi(dvf_tune("DVF_HEAD_DGRAD_MAXC")) : 32;   // tuning knob
    return dvf_head_applicable(d, nseg) && d->C_in <= maxc;
}

bool dvf_dconvt_applicable(const dvf_conv_desc *d, int nseg) {
    const bool k3 = d->KH == 3 && d->KW == 3, k4 = d->KH == 4 && d->KW == 4;
    return nseg == 1 && d->transposed && d->stride == 2 && d->pad == 1 && (k3 || k4) && d->C_out == 16 &&
           d->C_in >= 8 && d->C_in <= 64 && d->H_out <= 2 * d->H_in && d->W_out <= 2 * d->W_in &&
           (int64_t)d->H_in * d->W_in >= 64 * 64 && (int64_t)d->N * d->C_out * d->H_out * d->W_out < ((int64_t)1 << 31) &&
           dvf_tune("DVF_NO_DCONVT") == nullptr;
}

int dvf_dconvt_fwd(const dvf_conv_desc *d, const float *in, const float *w, const float *bias, float *out, hipStream_t st) {
    const dim3 grid(cdivh(d->W_in, 64), cdivh(d->H_in, 4), d->N);
    const size_t lds = (size_t)d->C_in * d->C_out * d->KH * d->KW * 4;
    if ((reinterpret_cast<uintptr_t>(w) & 15) != 0 || lds > 64 * 1024) return DVF_ERR_UNSUPPORTED;
#define DCONVT(CO, K) dconvt_s2_fwd_kernel<CO, K><<<grid, 256, lds, st>>>(in, w, bias, out, d->C_in, d->H_in, d->W_in, d->H_out, d->W_out, d->act, d->alpha, d->beta)
    if (d->KH == 3) DCONVT(16, 3);
    else DCONVT(16, 4);
#undef DCONVT
    DVF_LAUNCH_CHECK();
    dvf_plan_note(DVF_K_DCONVT_FWD, d->C_out, d->KH);
    return DVF_OK;
}

#define HEAD_DISPATCH(MOV, CALL)                 \
    switch (MOV) {                               \
        case 1: { constexpr int MO = 1; CALL; } break; \
        case 2: { constexpr int MO = 2; CALL; } break; \
        case 3: { constexpr int MO = 3; CALL; } break; \
        case 4: { constexpr int MO = 4; CALL; } break; \
        default: return DVF_ERR_UNSUPPORTED;     \
    }

#define HEAD_FWD_DISPATCH(MOV, CALL)             \
    switch (MOV) {                               \
        case 1: { constexpr int MO = 1; CALL; } break; \
        case 2: { constexpr int MO = 2; CALL; } break; \
        case 3: { constexpr int MO = 3; CALL; } break; \
        case 4: { constexpr int MO = 4; CALL; } break; \
        case 16: { constexpr int MO = 16; CALL; } break; \
        default: return DVF_ERR_UNSUPPORTED;     \
    }

// one forward-form launch (heads, thin full-resolution layers, narrow-segment dgrads): tile shape by resolution
static int head_fwd_launch(int MOv, const HeadSegs &in, const float *w, const float *bias, float *out, int C, int N, int H, int W,
                           int act, float alpha, float beta, int ws_m, int ws_c, int flip, const float *mask, hipStream_t st) {
    const bool lowres = (int64_t)N * H * W <= (int64_t)4 * 64 * 208 && MOv <= 4 && C >= 32;
    const int FTH = lowres ? 4 : FT_H, HCKT = lowres ? 32 : HCK;
    const int tilesX = cdivh(W, HT_W), tilesY = cdivh(H, FTH);
    const dim3 grid(tilesX * tilesY, N);
    const size_t lds = ((size_t)((MOv * C * 9 + 3) & ~3) + (size_t)HCKT * (FTH + 2) * HP_W) * 4;
#define HEAD_FWD_GO(HC, FH) head_fwd_kernel<MO, HC, FH><<<grid, 256, lds, st>>>(in, w, bias, out, C, H, W, tilesX, act, alpha, beta, ws_m, ws_c, flip, mask)
    if (lowres) {
        static bool big[5] = {false, false, false, false, false};
        HEAD_DISPATCH(MOv, {
            if (lds > 64 * 1024 && !big[MO]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&head_fwd_kernel<MO, 32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DVF_ERR_LAUNCH;
                big[MO] = true;
            }
            HEAD_FWD_GO(32, 4); });
    } else {
        HEAD_FWD_DISPATCH(MOv, (HEAD_FWD_GO(8, 8)));
    }
#undef HEAD_FWD_GO
    DVF_LAUNCH_CHECK();
    return DVF_OK;
}

int dvf_head_fwd(const dvf_conv_desc *d, const float *in, const float *w, const float *bias, float *out, hipStream_t st) {
    const int one = d->C_in;
    return dvf_head_fwd_segs(d, &in, &one, 1, w, bias, out, st);
}

// Thin full-resolution layers (iconv1: 16+1 -> 16 channels at the input resolution, DispNetS.py:126): a 32-row MFMA tile
// is half padding and the reduction (153 deep) too short to amortise a tile's prologue, so the direct kernel is ~2x faster.
bool dvf_head_wide_applicable(const dvf_conv_desc *d, int nseg) {
    return nseg >= 1 && nseg <= HEAD_MAX_SEGS && !d->transposed && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
           d->C_out == 16 && d->C_in <= 24 && d->H_out == d->H_in && d->W_out == d->W_in &&
           (int64_t)d->H_in * d->W_in >= 64 * 64 && dvf_tune("DVF_NO_HEAD") == nullptr && dvf_tune("DVF_NO_WIDE_HEAD") == nullptr;
}

int dvf_head_fwd_segs(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg, const float *w,
                      const float *bias, float *out, hipStream_t st) {
    HeadSegs in{};
    for (int s = 0; s < HEAD_MAX_SEGS; ++s) { in.p[s] = s < nseg ? in_segs[s] : in_segs[0]; in.c[s] = s < nseg ? seg_channels[s] : 0; }
    in.n = nseg;
    const int rc = head_fwd_launch(d->C_out, in, w, bias, out, d->C_in, d->N, d->H_in, d->W_in, d->act, d->alpha, d->beta,
                                   d->C_in * 9, 9, 0, nullptr, st);
    if (rc) return rc;
    dvf_plan_note(DVF_K_HEAD_FWD, d->C_out, nseg);
    return DVF_OK;
}

// dgrad of ONE narrow input segment (segc <= 4 channels starting at seg_off) of a 3x3 stride-1 pad-1 Conv2d:
//   din[n][ci][y][x] = sum_co sum_taps w[co][seg_off+ci][ta][tb] * dpre[n][co][y-ta+1][x-tb+1]
// = the head forward over dpre with transposed, tap-flipped weights.
bool dvf_head_seg_dgrad_applicable(const dvf_conv_desc *d, int segc) {
    const bool narrow = segc >= 1 && segc <= 4 && d->C_out >= 4 && d->C_out * segc <= 1024;
    const bool wide = segc == 16 && d->C_out <= 24 && (int64_t)d->H_in * d->W_in >= 64 * 64 &&     // see dvf_head_wide_applicable
                      dvf_tune("DVF_NO_WIDE_HEAD") == nullptr;
    return !d->transposed && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && (narrow || wide) &&
           d->H_out == d->H_in && d->W_out == d->W_in && dvf_tune("DVF_NO_HEAD") == nullptr;
}

int dvf_head_seg_dgrad(const dvf_conv_desc *d, const float *dpre, const float *w, float *din, int seg_off, int segc,
                       hipStream_t st, const float *mask) {
    const float *wseg = w + (int64_t)seg_off * 9;
    HeadSegs in{};
    for (int s = 0; s < HEAD_MAX_SEGS; ++s) { in.p[s] = dpre; in.c[s] = s == 0 ? d->C_out : 0; }
    in.n = 1;
    const int rc = head_fwd_launch(segc, in, wseg, nullptr, din, d->C_out, d->N, d->H_in, d->W_in, DVF_ACT_NONE, 1.f, 0.f, 9,
                                   d->C_in * 9, 1, mask, st);
    if (rc) return rc;
    dvf_plan_note(DVF_K_HEAD_SEG_DGRAD, segc);
    return DVF_OK;
}

int dvf_head_dgrad(const dvf_conv_desc *d, const float *dpre, const float *w, float *din, hipStream_t st, const float *mask) {
    const int tilesX = cdivh(d->W_in, HT_W), tilesY = cdivh(d->H_in, HT_H);
    const dim3 grid(tilesX * tilesY, d->N);
    HEAD_DISPATCH(d->C_out, (head_dgrad_kernel<MO><<<grid, 256, (size_t)d->C_in * ((MO * 9 + 3) & ~3) * 4, st>>>(
                                dpre, w, din, d->C_in, d->H_in, d->W_in, tilesX, mask)));
    DVF_LAUNCH_CHECK();
    dvf_plan_note(DVF_K_HEAD_DGRAD, d->C_out);
    return DVF_OK;
}

static void head_wgrad_grid(const dvf_conv_desc *d, int &tilesX, int &tilesY, int &chunks, int &groups) {
    tilesX = cdivh(d->W_in, HT_W); tilesY = cdivh(d->H_in, HT_H);
    chunks = cdivh(d->C_in, HWK);
    const int ntiles = d->N * tilesX * tilesY;
    groups = 2048 / chunks;
    if (groups < 1) groups = 1;
    if (groups > ntiles) groups = ntiles;
}

int64_t dvf_head_wgrad_ws_floats(const dvf_conv_desc *d) {
    int tilesX, tilesY, chunks, groups;
    head_wgrad_grid(d, tilesX, tilesY, chunks, groups);
    return (int64_t)groups * d->C_out * d->C_in * 9;
}

int dvf_head_wgrad(const dvf_conv_desc *d, const float *in, const float *dpre, float *dw, int accumulate, hipStream_t st,
                   float *ws, int64_t ws_floats) {
    if (!accumulate && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->C_out * d->C_in * 9, st) != hipSuccess)
        return DVF_ERR_LAUNCH;
    int tilesX, tilesY, chunks, groups;
    head_wgrad_grid(d, tilesX, tilesY, chunks, groups);
    const dim3 grid(chunks, groups);
    float *part = (ws && ws_floats >= dvf_head_wgrad_ws_floats(d)) ? ws : nullptr;
    HEAD_DISPATCH(d->C_out, (head_wgrad_kernel<MO><<<grid, 256, 0, st>>>(in, dpre, dw, d->N, d->C_in, d->H_in, d->W_in, tilesX,
                                                                        tilesY, part)));
    DVF_LAUNCH_CHECK();
    if (part) {
        const int n = d->C_out * d->C_in * 9;
        head_wgrad_finish_kernel<<<(n + 255) / 256, 256, 0, st>>>(part, dw, n, groups);
        DVF_LAUNCH_CHECK();
    }
    dvf_plan_note(DVF_K_HEAD_WGRAD, d->C_out, part ? 1 : 0);
    return DVF_OK;
}
