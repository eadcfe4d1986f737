// Shared helpers for the libdvf_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/dvf_hip.h"

#define DVF_LAUNCH_CHECK()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return DVF_ERR_LAUNCH; \
    } while (0)

static inline hipStream_t dvf_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Tuning knobs and ablation switches exist only in a -DDVF_TUNING build (`make TUNING=1`, used by tools/): the product
// library reads no environment variable on a launch path and carries no code that can switch parts of a kernel off.
#ifdef DVF_TUNING
static inline const char *dvf_tune(const char *name) { return getenv(name); }
#define DVF_DBG(a, bit) ((a).dbg & (bit))
#else
static inline const char *dvf_tune(const char *) { return nullptr; }
#define DVF_DBG(a, bit) (0)
#endif

// Plan log: every convolution entry point notes which kernel family and tiling it launched (thread-local, reset at the
// entry); dvf_conv2d_last_plans() reads it back.  Tests use it to prove that the plans exercised by the parity cases
// are the plans the benchmark step runs.
enum { DVF_K_PIPE = 1, DVF_K_GATHER = 2, DVF_K_HEAD_FWD = 3, DVF_K_HEAD_DGRAD = 4, DVF_K_HEAD_WGRAD = 5, DVF_K_WGRAD = 6,
       DVF_K_HEAD_SEG_DGRAD = 7, DVF_K_WGRAD_PIPE = 8, DVF_K_DCONVT_FWD = 9 };
constexpr int DVF_PLAN_INTS = 12, DVF_PLAN_MAX = 8;
struct DvfPlanLog { int n; int rec[DVF_PLAN_MAX][DVF_PLAN_INTS]; };
DvfPlanLog &dvf_plan_log();
static inline void dvf_plan_reset() { dvf_plan_log().n = 0; }
static inline void dvf_plan_note(int kernel, int a = 0, int b = 0, int c = 0, int d = 0, int e = 0, int f = 0, int g = 0,
                                 int h = 0, int i = 0, int j = 0, int k = 0) {
    DvfPlanLog &l = dvf_plan_log();
    if (l.n >= DVF_PLAN_MAX) return;
    const int v[DVF_PLAN_INTS] = {kernel, a, b, c, d, e, f, g, h, i, j, k};
    for (int x = 0; x < DVF_PLAN_INTS; ++x) l.rec[l.n][x] = v[x];
    ++l.n;
}

// Sum over the 64 lanes of a wave; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
