// Micro-benchmark: LDS accumulate throughput on gfx950 -- ds_add_f32 vs ds_add_u32 vs plain read-add-write, 64 consecutive
// dwords per wave-instruction (conflict-free), 4 or 8 waves per block, 1 block per CU.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, long long *clk) {
    __shared__ float t[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) t[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int idx = ((wave * 16 + c) * 64 + lane + it) & 8191;
            if (MODE == 0) atomicAdd(&t[idx], 1.0f);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(&t[idx]), 3u);
            else t[idx] += 1.0f;
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s += t[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *out; long long *clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    long long h[256];
    const char *names[3] = {"ds_add_f32", "ds_add_u32", "read+add+write"};
    for (int waves : {4, 8}) for (int mode = 0; mode < 3; ++mode) {
        const int iters = 200;
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k<0><<<256, waves * 64>>>(out, iters, clk);
            else if (mode == 1) k<1><<<256, waves * 64>>>(out, iters, clk);
            else k<2><<<256, waves * 64>>>(out, iters, clk);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, clk, 256 * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("%-16s %d waves/CU: %.1f clk per wave-instruction per CU (%.0f total)\n", names[mode], waves, avg / (iters * 16.0 * waves), avg);
    }
    return 0;
}
