// hipcc (ROCm 7.2) folds the float bitcast of the SECOND result of __builtin_amdgcn_permlane16_swap to the first one: k7 below compiles
// to `v_add_f32 v1, v1, v1` (k6, the same with integer adds, is right).  hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only
#include <hip/hip_runtime.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void row_swap(unsigned& x, unsigned& y) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
__global__ void k5(unsigned* p, unsigned* q) {
    unsigned x = p[threadIdx.x], y = q[threadIdx.x];
    row_swap(x, y);
    p[threadIdx.x] = x;
    q[threadIdx.x] = y;
}
__global__ void k6(unsigned* p, unsigned* q) {
    unsigned x = p[threadIdx.x], y = q[threadIdx.x];
    const u32x2 r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    p[threadIdx.x] = r[0] + r[1];
}
__global__ void k7(float* p, float* q) {
    float x = p[threadIdx.x], y = q[threadIdx.x];
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    p[threadIdx.x] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
