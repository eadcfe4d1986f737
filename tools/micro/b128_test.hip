#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFF0, 0x00020000);
}
__global__ void probe(const float *src, float *out) {
    const int lane = threadIdx.x;
    auto r = rsrc(src);
    const int seg = lane >> 3, q = lane & 7;
    const unsigned lane_off = (unsigned)seg * 4000u + ((unsigned)(4 * q) << 2);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (lane == 5) ? 0x80000000u : lane_off, 160u, 0));
    out[lane * 4 + 0] = v4.x; out[lane * 4 + 1] = v4.y; out[lane * 4 + 2] = v4.z; out[lane * 4 + 3] = v4.w;
}
int main() {
    float *src, *out;
    std::vector<float> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = (float)i;
    (void)hipMalloc(&src, 16384 * 4); (void)hipMalloc(&out, 256 * 4);
    (void)hipMemcpy(src, h.data(), 16384 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, out);
    std::vector<float> o(256);
    (void)hipMemcpy(o.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        const float exp = (l == 5) ? 0.f : (float)(40 + (l >> 3) * 1000 + 4 * (l & 7) + j);
        if (o[l * 4 + j] != exp) { if (bad < 8) printf("lane %d j %d got %g exp %g\n", l, j, o[l * 4 + j], exp); ++bad; }
    }
    printf("bad %d\n", bad);
    return 0;
}
