// How fast can the producer waves of one CU issue LDS-DMA (buffer_load ... lds) beside four MFMA waves?
// One 512-thread block per CU: waves 0-3 run the conv_pipe MFMA pattern (optional), waves 4..4+P-1 issue DMA pieces in
// batches of B instructions followed by s_waitcnt vmcnt(0).  Source: a window of `span` bytes per block (L2-hot when small).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -o dma_rate dma_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFF0, 0x00020000);
}

// LB: bytes per lane (4 / 16); P: producer waves; MF: MFMA waves busy; PRIO: s_setprio of the producers; ROWS: lanes of a
// piece walk rows of 40 floats (patch-like: 10 lanes x 16 B or 40 x 4 B per row, row pitch 416 floats) instead of 1 KiB runs
template <int LB, int P, int MF, int PRIO, int ROWS>
__global__ __launch_bounds__(512) void k(const float *src, float *out, unsigned long long *st, int batches, int B, unsigned span) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave >= 4) {
        const int pidx = wave - 4;
        if (pidx >= P) return;
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        const __amdgpu_buffer_rsrc_t r = rsrc(src);
        unsigned voff;
        if (ROWS) {
            const int fl = LB == 16 ? lane * 4 : lane;                // float index inside the piece
            voff = (unsigned)((fl / 40) * 416 + (fl % 40)) << 2;
        } else {
            voff = (unsigned)lane * LB;
        }
        unsigned long long t0, t1, iss = 0, wt = 0, t2;
        const unsigned base = ((unsigned)blockIdx.x * 65536u + (unsigned)pidx * 16384u) % span;
        unsigned soff = base;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        const unsigned long long tstart = t0;
        for (int b = 0; b < batches; ++b) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int i = 0; i < B; ++i) {
                float *dst = lds + 16384 + pidx * 4096 + (i & 7) * (LB == 16 ? 256 : 64);
                if constexpr (LB == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)dst, 16, voff, soff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)dst, 4, voff, soff, 0, 0);
                soff += ROWS ? 416 * 4 * 8 : 1024;
                if (soff >= span) soff -= span;
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
            iss += t1 - t0;
            wt += t2 - t1;
        }
        if (lane == 0) {
            st[(blockIdx.x * 4 + pidx) * 3 + 0] = iss;
            st[(blockIdx.x * 4 + pidx) * 3 + 1] = wt;
            st[(blockIdx.x * 4 + pidx) * 3 + 2] = t2 - tstart;
        }
        return;
    }
    if (!MF) return;
    // MFMA waves: ds_read + MFMA like the conv_pipe consumer (2 x b128 + 4 x b32 per 8 MFMAs), roughly as long as the producers run
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float *ap = lds + lane * 4, *bp = lds + 8192 + (lane & 31);
    const int iters = batches * B * (LB == 16 ? 6 : 2) / (8 * P) + 8;
    for (int it = 0; it < iters; ++it) {
        const float4 a0 = *reinterpret_cast<const float4 *>(ap + (it & 7) * 256), a1 = *reinterpret_cast<const float4 *>(ap + (it & 7) * 256 + 2048);
        const float b0 = bp[it & 63], b1 = bp[64 + (it & 63)], b2 = bp[128 + (it & 63)], b3 = bp[192 + (it & 63)];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b3, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b3, acc1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int LB, int P, int MF, int PRIO, int ROWS> void run(const char *tag, const float *src, unsigned span, int B) {
    const int blocks = 256, batches = 40;
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&st, blocks * 4 * 3 * 8);
    (void)hipMemset(st, 0, blocks * 4 * 3 * 8);
    auto fn = k<LB, P, MF, PRIO, ROWS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    fn<<<blocks, 512, 144 * 1024>>>(src, out, st, batches, B, span);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    fn<<<blocks, 512, 144 * 1024>>>(src, out, st, batches, B, span);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 12);
    (void)hipMemcpy(h.data(), st, blocks * 12 * 8, hipMemcpyDeviceToHost);
    std::vector<double> iss, wt, tot;
    for (int b = 0; b < blocks; ++b) { iss.push_back((double)h[b * 12]); wt.push_back((double)h[b * 12 + 1]); tot.push_back((double)h[b * 12 + 2]); }
    std::sort(iss.begin(), iss.end()); std::sort(wt.begin(), wt.end()); std::sort(tot.begin(), tot.end());
    const double n = (double)batches * B;
    const double bytes_cu = n * P * 64.0 * LB;
    printf("%-52s B %2d | issue %6.0f cyc/instr, drain %6.0f cyc/batch | %5.1f instr/us/CU, %5.1f GB/s/CU (%5.2f TB/s chip) | kernel %.0f us\n", tag, B,
           iss[blocks / 2] / n, wt[blocks / 2] / batches, n * P / (tot[blocks / 2] / 2.3e3), bytes_cu / (tot[blocks / 2] / 2.3), bytes_cu * 256 / (tot[blocks / 2] / 2.3) / 1e3, ms * 1e3);
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    float *src;
    const size_t bytes = 512u << 20;
    (void)hipMalloc(&src, bytes + (1u << 20));
    (void)hipMemset(src, 0, bytes);
    for (unsigned span : {8u << 20, 512u << 20}) {
        printf("--- source window %u MiB\n", span >> 20);
        run<16, 1, 0, 0, 0>("16 B lanes, 1 producer, no MFMA", src, span, 16);
        run<16, 2, 0, 0, 0>("16 B lanes, 2 producers, no MFMA", src, span, 16);
        run<16, 4, 0, 0, 0>("16 B lanes, 4 producers, no MFMA", src, span, 16);
        run<16, 2, 1, 0, 0>("16 B lanes, 2 producers, MFMA waves busy", src, span, 16);
        run<16, 4, 1, 0, 0>("16 B lanes, 4 producers, MFMA waves busy", src, span, 16);
        run<16, 4, 1, 3, 0>("16 B lanes, 4 producers prio 3, MFMA busy", src, span, 16);
        run<16, 4, 1, 0, 0>("16 B lanes, 4 producers, MFMA busy", src, span, 4);
        run<4, 2, 1, 0, 0>("4 B lanes, 2 producers, MFMA waves busy", src, span, 16);
        run<4, 4, 1, 0, 0>("4 B lanes, 4 producers, MFMA waves busy", src, span, 16);
        run<4, 4, 1, 3, 0>("4 B lanes, 4 producers prio 3, MFMA busy", src, span, 16);
        run<16, 4, 1, 0, 1>("16 B lanes, patch rows, 4 producers, MFMA busy", src, span, 16);
        run<4, 4, 1, 0, 1>("4 B lanes, patch rows, 4 producers, MFMA busy", src, span, 16);
        run<16, 2, 1, 0, 1>("16 B lanes, patch rows, 2 producers, MFMA busy", src, span, 16);
        run<4, 2, 1, 0, 1>("4 B lanes, patch rows, 2 producers, MFMA busy", src, span, 16);
    }
    return 0;
}
