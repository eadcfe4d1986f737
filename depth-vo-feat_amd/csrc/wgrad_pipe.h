// Pipelined weight-gradient kernel (wgrad_pipe.hip), shared with the planner in conv.hip.
//
//   G[m][c][ta][tb] += sum over (n, y, x) of  P[n][m][y][x] * Q[n][c][y*S + ta - pad][x*S + tb - pad]
//
// Conv2d: P = dL/dpre, Q = layer input;  ConvTranspose2d: P = layer input, Q = dL/dpre (conv.hip fills the roles).
#pragma once
#include "dvf_common.h"

struct WgpArgs {
    const float *P;      // [N, PCtot, GH, GW]; channels m_base .. m_base+M-1 are used
    const float *Q[DVF_MAX_SEGS];   // virtual concatenation of nseg tensors [N, segC[s], QH, QW]: Cq = sum of segC
    int segC[DVF_MAX_SEGS];
    int nseg;
    float *G;            // G[(g_mbase + m) * g_mstride + (g_cbase + c) * KK + tap]
    int PCtot, m_base, M, Cq;
    int64_t g_mstride;
    int g_mbase, g_cbase, KK, KH, KW;
    int N, GH, GW, QH, QW, S, pad;
    int CK;              // Q channels per column chunk (CK * KK <= 4 * NTW * tile columns)
    int tilesX, tilesY, ntiles, mtiles, cchunks;
    int PHq, RSq, PSq, NPIq, XA;   // Q patch: rows, row stride, channel stride (floats), DMA pieces per channel, aligned left margin
    int x4;              // 1: 16-byte DMA lanes (GW % 4 == 0, QW % 4 == 0, 16-byte aligned bases), 0: 4-byte lanes
    int W;               // work items = mtiles * cchunks * ntiles, split evenly over the blocks of the launch
    float *dbias;        // bias gradient of a Conv2d layer (P = dL/dpre): dbias[m] += sum over pixels of P[m], or NULL
    int bias_col;        // column of the block's tile that multiplies P by a slot of ones (-1: none); QSLOTS = CK + 1 then
    int QSLOTS;          // channel slots of PSq floats per LDS stage
    // Deterministic flush (ws != NULL): a block stores the accumulator set it holds when its item range leaves output
    // block mc as one plain, coalesced slab, ws[(block + mc) * slot .. ) -- (block + mc) is unique per (block, mc) pair
    // because the ranges are contiguous and ordered -- and wgrad_reduce_kernel adds the slabs of every output block in
    // block order into G / dbias.  ws == NULL: float atomics straight into G (order of arrival).
    float *ws;
    int dbg;             // ablation switches (-DDVF_TUNING builds): 1 no DMA loads, 4 no MFMA, 8 no atomic epilogue
    unsigned long long *stamps;   // -DDVF_TUNING builds: per-block cycle account (8 x u64 per block), or NULL
};

constexpr int WGP_BH = 4, WGP_BW = 32;     // pixel tile of one pipeline step: 4 rows x 32 columns of the P grid
constexpr int WGP_PAIR = 260;              // LDS floats per pair of P channels (2 x 128 pixels + 4: bank skew)
constexpr int WGP_MAXQ = 16;               // DMA pieces per Q patch channel
constexpr size_t WGP_LDS_CAP = 160 * 1024;

// (MT, NTW) in {1,2}^2; S in {1,2}.  Returns DVF_OK / DVF_ERR_*.
// tile = 32: 32*MT rows x 128*NTW columns per block (NTW 1..2); tile = 16: 16*MT rows x 64*NTW columns (NTW 1..4)
int dvf_wgrad_pipe_launch(const WgpArgs &a, int MT, int NTW, int nblocks, size_t lds_bytes, hipStream_t st, int tile = 32);
int dvf_wgrad_pipe_reduce(const WgpArgs &a, int MT, int NTW, int nblocks, hipStream_t st, int tile);
// floats of one flush slab of the (MT, NTW, tile) variant, and the number of slabs a launch of `nblocks` blocks may write
inline int64_t wgrad_pipe_slot_floats(int MT, int NTW, int tile) { return (int64_t)tile * MT * 4 * NTW * tile; }
inline int64_t wgrad_pipe_slots(const WgpArgs &a, int nblocks) { return (int64_t)nblocks + (int64_t)a.mtiles * a.cchunks; }
