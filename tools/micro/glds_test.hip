// Semantics probe for buffer_load ... lds on gfx950: OOB lanes, exec-masked lanes, 16-byte width.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFF0, 0x00020000);
}
__global__ void probe(const float *src, float *out) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = -7.f;
    __syncthreads();
    auto r = rsrc(src);
    // test 1: dword, lanes 10..19 OOB, lanes >= 40 exec-masked
    if (lane < 40) {
        unsigned voff = (lane >= 10 && lane < 20) ? 0x80000000u : (unsigned)lane * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)&lds[0], 4, voff, 0, 0, 0);
    }
    // test 2: 16 B per lane with soffset, lanes 4..5 OOB, into lds[256..511]
    {
        unsigned voff = (lane == 4 || lane == 5) ? 0x80000000u : (unsigned)lane * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)&lds[256], 16, voff, 400, 0, 0);
    }
    // test 3: permuted source (lane l reads element 63-l), into lds[512..575]
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)&lds[512], 4, (unsigned)(63 - lane) * 4u, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    float *src, *out;
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i + 1.f;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 1024 * 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, out);
    std::vector<float> o(1024);
    hipMemcpy(o.data(), out, 1024 * 4, hipMemcpyDeviceToHost);
    printf("t1:"); for (int i = 0; i < 64; ++i) printf(" %g", o[i]); printf("\n");
    printf("t2:"); for (int i = 256; i < 256 + 40; ++i) printf(" %g", o[i]); printf(" ... %g %g\n", o[510], o[511]);
    printf("t3:"); for (int i = 512; i < 520; ++i) printf(" %g", o[i]); printf(" ... %g\n", o[575]);
    return 0;
}
