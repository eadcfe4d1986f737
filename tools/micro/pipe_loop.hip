// The consumer loop of conv_pipe_kernel<2,1,1,8,3> in isolation: what does ONE fp32 MFMA cost (cycles, and at which
// clock) when its fragments come from LDS the way the kernel reads them?  4 MFMA waves (one per SIMD) + 4 stand-in
// producer waves that only join the chunk barriers.  Stamps: s_memtime (shader cycles) and s_memrealtime (100 MHz).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -o pipe_loop pipe_loop.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE bits: 1 A reads (2 x ds_read_b128 per tap), 2 B reads (12 x ds_read_b32 per unit), 4 round-2 order (reads behind the
// LAST MFMA group of a tap), 8 one barrier per chunk (with the producer stand-ins), 16 two MFMA waves per SIMD (all 8 waves compute)
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *st, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) lds[i] = (float)((i * 2654435761u) >> 20) * (1.0f / 4096.f) - 0.5f;
    __syncthreads();
    if (wave >= 4 && !(MODE & 16)) {
        if (MODE & 8)
            for (int c = 0; c < chunks; ++c) __builtin_amdgcn_s_barrier();
        return;
    }
    f32x16 acc[2];
    for (int m = 0; m < 2; ++m)
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float *ap = lds + lane * 4;                                  // A: [unit*3+tap][m][lane][4]
    const float *bp = lds + 8192 + (lane & 31) + (lane >> 5) * 328;    // B: rows of a patch
    f4 af[2][2];
    float bf[2][3][4];
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    auto load_a = [&](auto tpc, int idx) {
        constexpr int tp = decltype(tpc)::value;
        if (MODE & 1) {
            af[tp][0] = *reinterpret_cast<const f4 *>(ap + (idx & 15) * 512);
            af[tp][1] = *reinterpret_cast<const f4 *>(ap + (idx & 15) * 512 + 256);
        }
    };
    auto load_bj = [&](auto bufc, auto jc, int unit) {
        constexpr int buf = decltype(bufc)::value, j = decltype(jc)::value;
        if (MODE & 2) {
            const float *p = bp + (unit & 3) * 34 + j * 656;
#pragma unroll
            for (int u = 0; u < 3; ++u) bf[buf][u][j] = p[u];
        }
    };
    auto mma_j = [&](auto bufc, auto tpc, auto uc, auto jc) {
        constexpr int buf = decltype(bufc)::value, tp = decltype(tpc)::value, u = decltype(uc)::value, j = decltype(jc)::value;
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tp][m][j], bf[buf][u][j], acc[m], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    int lk = 0;
    auto step = [&](auto bufc, auto nbufc, auto pbc, auto uc) {
        constexpr int u = decltype(uc)::value, tp = (decltype(pbc)::value + u) & 1;
        using TP = std::integral_constant<int, tp>;
        using NTP = std::integral_constant<int, tp ^ 1>;
        if constexpr (MODE & 4) {
            mma_j(bufc, TP{}, uc, I0{}); mma_j(bufc, TP{}, uc, I1{}); mma_j(bufc, TP{}, uc, I2{}); mma_j(bufc, TP{}, uc, I3{});
            if constexpr (u == 0) { load_bj(nbufc, I0{}, lk); load_bj(nbufc, I1{}, lk); load_bj(nbufc, I2{}, lk); load_bj(nbufc, I3{}, lk); }
            load_a(NTP{}, lk * 3 + u + 1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            mma_j(bufc, TP{}, uc, I0{});
            __builtin_amdgcn_sched_barrier(0);
            load_a(NTP{}, lk * 3 + u + 1);
            if constexpr (u == 0) load_bj(nbufc, I0{}, lk);
            __builtin_amdgcn_sched_barrier(0);
            mma_j(bufc, TP{}, uc, I1{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u == 0) { load_bj(nbufc, I1{}, lk); __builtin_amdgcn_sched_barrier(0); }
            mma_j(bufc, TP{}, uc, I2{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u == 0) { load_bj(nbufc, I2{}, lk); __builtin_amdgcn_sched_barrier(0); }
            mma_j(bufc, TP{}, uc, I3{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u == 0) { load_bj(nbufc, I3{}, lk); __builtin_amdgcn_sched_barrier(0); }
        }
    };
    auto unit = [&](auto bufc, auto nbufc, auto pbc) {
        step(bufc, nbufc, pbc, I0{});
        step(bufc, nbufc, pbc, I1{});
        step(bufc, nbufc, pbc, I2{});
        ++lk;
    };
    // registers hold something even when a read class is switched off
    for (int tp = 0; tp < 2; ++tp)
        for (int m = 0; m < 2; ++m) af[tp][m] = f4{0.25f + lane, 0.5f, 0.75f, 1.f};
    for (int b = 0; b < 2; ++b)
        for (int u = 0; u < 3; ++u)
            for (int j = 0; j < 4; ++j) bf[b][u][j] = 0.125f * (u + j + 1);
    for (int c = 0; c < chunks; ++c) {
        if (MODE & 8) __builtin_amdgcn_s_barrier();
        lk = 0;
        load_bj(I0{}, I0{}, 0); load_bj(I0{}, I1{}, 0); load_bj(I0{}, I2{}, 0); load_bj(I0{}, I3{}, 0);
        load_a(I0{}, 0);
        __builtin_amdgcn_sched_barrier(0);
        for (int kk = 0; kk < 6; kk += 2) {            // 6 units x 3 taps x 8 MFMAs = 144 per chunk
            unit(I0{}, I1{}, I0{});
            unit(I1{}, I0{}, I1{});
        }
    }
    asm volatile("s_nop 0" : "+v"(acc[0]), "+v"(acc[1]));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0 && wave == 0) {
        st[2 * blockIdx.x] = t1 - t0;
        st[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int MODE> void run(const char *tag, int blocks, int chunks) {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    (void)hipMalloc(&st, (size_t)blocks * 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<MODE><<<blocks, 512, 80 * 1024>>>(out, st, chunks);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 512, 80 * 1024>>>(out, st, chunks);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), st, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[2 * b]); clk.push_back((double)h[2 * b] / ((double)h[2 * b + 1] * 10.0)); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double nm = 144.0 * chunks;
    const int waves = (MODE & 16) ? 8 : 4;
    printf("%-58s blocks %3d chunks %3d | %7.1f us wall | loop %8.0f cyc = %5.1f cyc/MFMA (max %5.1f) | clock %.2f GHz | %5.1f TF\n", tag, blocks,
           chunks, best * 1e3, cyc[blocks / 2], cyc[blocks / 2] / nm, cyc[blocks - 1] / nm, clk[blocks / 2],
           (double)blocks * waves * nm * 4096.0 / (best * 1e-3) / 1e12);
    (void)hipFree(out); (void)hipFree(st);
}
int main() {
    for (int chunks : {8, 80}) {
        run<0>("MFMA only (operands in registers)", 256, chunks);
        run<1>("+ A reads, new order", 256, chunks);
        run<3>("+ A + B reads, new order", 256, chunks);
        run<7>("+ A + B reads, round-2 order", 256, chunks);
        run<11>("+ A + B reads, new order, chunk barriers", 256, chunks);
        run<15>("+ A + B reads, round-2 order, chunk barriers", 256, chunks);
        run<19>("+ A + B reads, new order, 2 MFMA waves per SIMD", 256, chunks);
        run<11>("+ A + B reads, new order, chunk barriers", 224, chunks);
    }
    return 0;
}
