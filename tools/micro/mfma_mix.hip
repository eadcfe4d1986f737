// What may sit between two fp32 MFMAs of one wave without slowing the matrix pipe?  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float x = a + threadIdx.x, y = b;
    int s0 = blockIdx.x, s1 = 3, s2 = 5, s3 = 7;
    int v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    unsigned laddr = (threadIdx.x & 63) * 4;
    float l0 = 0, l1 = 0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f3 __attribute__((ext_vector_type(3)));
    f2 l2 = {0, 0}; f3 l3 = {0, 0, 0};
    unsigned laddr4 = (threadIdx.x & 63) * 4 + 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            MF(acc0);
            if (MODE & 1) asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
            if (MODE & 2) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %1, %1, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (MODE & 4) asm volatile("ds_read_b32 %0, %1" : "=v"(l0) : "v"(laddr));
            if (MODE & 16) asm volatile("s_cmp_lt_i32 %0, 100000000\n s_cselect_b32 %1, %2, %3" : "+s"(s0), "+s"(s1) : "s"(s2), "s"(s3) : "scc");
            if (MODE & 32) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0));
            if (MODE & 64) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));
            if (MODE & 128) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));
            if (MODE & 256) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(l2) : "v"(laddr));
            if (MODE & 512) asm volatile("ds_read_b96 %0, %1" : "=v"(l3) : "v"(laddr4));
            MF(acc1);
            if (MODE & 32) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0));
            if (MODE & 64) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));
            if (MODE & 128) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));
            if (MODE & 256) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(l2) : "v"(laddr));
            if (MODE & 512) asm volatile("ds_read_b96 %0, %1" : "=v"(l3) : "v"(laddr4));
            if (MODE & 1) asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
            if (MODE & 2) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %1, %1, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (MODE & 4) asm volatile("ds_read_b32 %0, %1" : "=v"(l1) : "v"(laddr));
            if (MODE & 8) asm volatile("s_waitcnt lgkmcnt(1)");
        }
        if (MODE & (4 | 256 | 512)) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = l0 + l1 + v0 + v1 + v2 + s0 + s1 + s2 + l2[0] + l2[1] + l3[0] + l3[1] + l3[2];
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *tag) {
    const int blocks = 256, iters = 2000;
    float *out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, 1.f, 2.f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("%-44s %.3f ms  %.1f TFLOP/s\n", tag, best, flops / best / 1e9);
    (void)hipFree(out);
}
int main() {
    run<0>("mfma only");
    run<1>("+6 SALU per mfma");
    run<2>("+3 VALU per mfma");
    run<3>("+6 SALU +3 VALU");
    run<4>("+1 ds_read_b32 per mfma (no waits)");
    run<12>("+1 ds_read_b32 per mfma, lgkmcnt(1) per pair");
    run<7>("+6 SALU +3 VALU +1 ds_read");
    run<16>("+cmp/cselect per pair");
    run<31>("everything");
    run<32>("+1 SALU per mfma");
    run<64>("+1 VALU per mfma");
    run<128>("+2 VALU (v_add3-like dependent chain) per mfma");
    run<256>("+1 ds_read2_b32 per mfma");
    run<512>("+1 misaligned ds_read_b96 per mfma");
    return 0;
}
