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

// Sum over the 64 lanes of a wave; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
