// Sustained v_mfma_f32_32x32x2_f32 rate: NACC independent accumulators per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, const char *tag) {
    float *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
        printf("%s NACC=%d blocks=%d: %.3f ms  %.1f TFLOP/s\n", tag, NACC, blocks, ms, flops / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    run<1>(256, "1 wave/SIMD");
    run<2>(256, "1 wave/SIMD");
    run<4>(256, "1 wave/SIMD");
    run<2>(512, "2 waves/SIMD");
    run<2>(1024, "4 waves/SIMD");
    return 0;
}
