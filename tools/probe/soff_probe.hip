// Is the scalar offset of a raw buffer load part of the range check against num_records?  (gfx950)
// out[0]: voffset in range, soffset pushes the address past num_records      out[1]: voffset itself past num_records
// out[2]: everything in range.  The buffer holds 1.0f everywhere (4096 floats, the descriptor covers the first 256 bytes).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, float* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 256, 0x00020000);
    if (threadIdx.x == 0) {
        out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, 1024, 0));
        out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 1024, 0, 0));
        out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 16, 16, 0));
        out[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 128, 192, 0));
    }
}
int main() {
    float *src, *out, h[4096], o[4];
    for (int i = 0; i < 4096; ++i) h[i] = 1.0f;
    hipMalloc(&src, sizeof(h)); hipMalloc(&out, 16);
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
    hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("soffset past num_records (voffset in range): %g   voffset past: %g   in range: %g   voffset in range, voffset+soffset past: %g\n", o[0], o[1], o[2], o[3]);
    printf("=> the scalar offset %s part of the range check\n", o[0] == 0.f ? "IS" : "is NOT");
    return 0;
}
