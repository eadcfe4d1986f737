// What does a launch cost before any work?  Back-to-back launches of do-nothing kernels of the shapes the convolution
// kernels use (256 blocks; 256 / 384 / 512 threads; 0 / 64 / 125 / 147 KB of dynamic LDS; a few barriers), timed with
// HIP events over 200 launches.   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { int v[100]; };
__global__ void k_empty(Big a, float *out) {
    if (a.v[0] == 12345) out[threadIdx.x] = 1.f;
}
__global__ void k_lds(Big a, float *out) {
    extern __shared__ float sm[];
    if (a.v[0] == 12345) { sm[threadIdx.x] = 1.f; out[threadIdx.x] = sm[threadIdx.x ^ 1]; }
}
__global__ void k_bar(Big a, float *out) {
    extern __shared__ float sm[];
    for (int i = 0; i < a.v[1]; ++i) __syncthreads();
    if (a.v[0] == 12345) { sm[threadIdx.x] = 1.f; out[threadIdx.x] = sm[threadIdx.x ^ 1]; }
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 200.f;
}
int main() {
    float *out; hipMalloc(&out, 4096);
    Big a{}; a.v[1] = 5;
    hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_bar, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int blocks[] = {64, 256, 512, 1024};
    const int thr[] = {256, 512};
    const int lds[] = {0, 64 * 1024, 125 * 1024, 147 * 1024};
    for (int b : blocks) for (int t : thr) {
        printf("blocks %4d threads %3d | empty %.2f us", b, t, timeit([&] { k_empty<<<b, t>>>(a, out); }));
        for (int l : lds) printf(" | lds %3dK %.2f", l / 1024, timeit([&] { k_lds<<<b, t, l>>>(a, out); }));
        printf(" | 5 barriers lds 125K %.2f\n", timeit([&] { k_bar<<<b, t, 125 * 1024>>>(a, out); }));
    }
    return 0;
}
