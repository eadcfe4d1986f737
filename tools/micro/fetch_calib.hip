// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this project uses (MI355X_MICROARCH.md: FETCH_SIZE
// reads exactly 1/2 of the bytes of a 16-B-per-lane stream; "other access widths are uncalibrated").  Each kernel reads a
// 1 GiB buffer (far beyond the 256 MiB Infinity Cache) exactly once with a known width:
//   read16   global_load_dwordx4, one per lane, coalesced            (act_bwd / adam / tile staging)
//   read4    global_load_dword, one per lane, coalesced              (row-strided patch loads of the unpacked conv kernels)
//   read8u   global_load_dwordx2 at 4-byte-aligned (odd) positions   (the 2-tap "pair" gathers of the warp kernels)
//   dma4     buffer_load_dword ... lds, 64 consecutive floats        (LDS-DMA patch pieces of conv_pipe_kernel)
//   dma16    buffer_load_dwordx4 ... lds, 1 KiB per instruction      (LDS-DMA packed-weight pieces)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// and divide each kernel's FETCH_SIZE (KiB) * 1024 by the bytes printed here.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr size_t NF = (size_t)1 << 28;      // floats = 1 GiB

__global__ void read16(const f4 *p, float *out, size_t n4) {
    f4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s.x + s.y + s.z + s.w == 1.2345f) out[0] = s.x;
}
__global__ void read4(const float *p, float *out, size_t n) {
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 1.2345f) out[0] = s;
}
struct __attribute__((packed, aligned(4))) Pair { float x, y; };
__global__ void read8u(const float *p, float *out, size_t n2) {   // pairs starting at odd float indices: 8 bytes, 4-byte aligned
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2 - 1; i += (size_t)gridDim.x * blockDim.x) {
        const Pair v = *reinterpret_cast<const Pair *>(p + 2 * i + 1);
        s += v.x + v.y;
    }
    if (s == 1.2345f) out[0] = s;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFF0, 0x00020000);
}
template <int BYTES>
__global__ void dma(const float *p, float *out, size_t nfloats) {
    __shared__ __attribute__((aligned(16))) float l[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = 64 * BYTES / 4;                    // floats per wave-instruction
    const size_t nchunk = nfloats / PER;
    float s = 0;
    for (size_t c = (size_t)blockIdx.x * 4 + wave; c < nchunk; c += (size_t)gridDim.x * 4) {
        const size_t base = c * PER;                       // floats; split into a 2 GiB-safe descriptor base + 32-bit offset
        const __amdgpu_buffer_rsrc_t r = rsrc(p + (base & ~(size_t)0xFFFFFF));
        const unsigned off = (unsigned)((base & 0xFFFFFF) * 4);
        if constexpr (BYTES == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)&l[wave][0], 4, (unsigned)lane * 4, off, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)&l[wave][0], 16, (unsigned)lane * 16, off, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s += l[wave][lane];
    }
    if (s == 1.2345f) out[0] = s;
}
int main() {
    float *buf, *out;
    hipMalloc(&buf, NF * 4); hipMalloc(&out, 64);
    hipMemset(buf, 0, NF * 4);
    hipDeviceSynchronize();
    read16<<<2048, 256>>>((const f4 *)buf, out, NF / 4);
    read4<<<2048, 256>>>(buf, out, NF);
    read8u<<<2048, 256>>>(buf, out, NF / 2);
    dma<4><<<1024, 256>>>(buf, out, NF);
    dma<16><<<1024, 256>>>(buf, out, NF);
    hipDeviceSynchronize();
    printf("bytes read by each kernel: %zu\n", NF * 4);
    return 0;
}
